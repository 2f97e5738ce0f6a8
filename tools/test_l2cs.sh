#!/bin/bash
# The reference's tools/test_l2cs.sh (L2CS setting: frames up to 448x448, no crop, padded to multiples of 32).  The config is the
# reference's own file; this repo's equivalent is configs/mcgaze/r50_clip7_l2cs.py.
python tools/test_gaze360_gaze.py configs/multiclue_gaze/multiclue_gaze_r50_l2cs.py ckpts/multiclue_gaze_r50_l2cs.pth --json data/l2cs/test.json --root data/l2cs/test_rawframes/
python tools/calculate_mae_l2cs.py --evalfile results/results_multiclue_gaze_r50_l2cs_test.json
