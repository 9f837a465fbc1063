export E2E_LONG=16 E2E_QUIET=1
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  echo "GPU_MAX_HW_QUEUES=$q"
  for cfg in "4 1024 208 1 small 2 host_u8" "4 1024 256 1 small 2 host_u8" "4 1024 208 1 small 2 host_f32" "4 1024 256 1 small 2 resident" "4 1024 208 1 small 1 host_u8"; do
    timeout -k 10 120 python tools/e2e_timeline.py $cfg 2>&1 | grep unrecorded
  done
done
