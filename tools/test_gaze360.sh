#!/bin/bash
# The reference's tools/test_gaze360.sh: dataset inference, then the MAE.  The config is the reference's own file (it loads
# unchanged); this repo's equivalent is configs/mcgaze/r50_clip7_gaze360.py.  One GPU; for N GPUs prefix the first line with
# `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1` (videos are sharded over the ranks).
python tools/test_gaze360_gaze.py configs/multiclue_gaze/multiclue_gaze_r50_gaze360.py ckpts/multiclue_gaze_r50_gaze360.pth --json data/gaze360/test.json --root data/gaze360/test_rawframes/
python tools/calculate_mae_gaze360.py --evalfile results/results_multiclue_gaze_r50_gaze360_test.json
