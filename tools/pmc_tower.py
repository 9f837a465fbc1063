#!/usr/bin/env python3
"""A few eager forwards of a tower for a PMC pass (bash tools/pmc_cmd.sh "tools/pmc_tower.py" FETCH_SIZE WRITE_SIZE): which kernel moves more than its algorithm needs.
FETCH_SIZE / WRITE_SIZE come in KB; FETCH is doubled on gfx950 (MI355X_MICROARCH.md).  python tools/pmc_tower.py [vit|text] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import clip_text, clip_vit  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "vit"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
with torch.no_grad():
	if which == "vit":
		t = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).cuda()
		x = torch.randn(B, 3, 224, 224).cuda()
	else:
		t = clip_text.NativeTextTower(clip_text.TEXT_B_32, seed=3).cuda()
		x = torch.randint(1, 49000, (B, 77)).cuda()
		x[:, -1] = 49407
	t.use_graphs = False
	for _ in range(4):
		t(x)
	torch.cuda.synchronize()
