#!/bin/bash
# Long runs (16 x 12 batches of 256) of the coalesced ViT-B/32 + greedy pipeline: tools/e2e_timeline.py [batches per launch] [decode rows] [budget] [ahead] [small|big] [decode lanes] [source]
export E2E_LONG=${E2E_LONG:-16} E2E_QUIET=1
for cfg in "4 1024 208 1 small 1 resident" "4 1024 256 1 small 2 resident" "4 1024 208 1 small 1 host_u8" "4 1024 208 1 small 2 host_u8" "4 1024 256 1 small 2 host_u8" "4 1024 208 1 small 1 host_f32" "4 1024 208 1 small 2 host_f32"; do
  timeout -k 10 120 python tools/e2e_timeline.py $cfg 2>&1 | grep unrecorded
done
