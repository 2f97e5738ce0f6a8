"""L2CS mean angular error of a result file -- the reference's command line (tools/calculate_mae_l2cs.py:7-14:
``--evalfile results/results_<cfg>_<json> [--anno data/l2cs/test.json]``; annotation index ``anno_id * 3`` :110, front-20 also
needs |pitch| <= 20 degrees :132-139; prints the three ``fusion_gazes`` lines).  The arithmetic is mcgaze_amd.metric.gaze_error."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcgaze_amd import metric  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='L2CS MAE of a result json')
    parser.add_argument('--evalfile', help='pred_gaze json file', default='results/results_multiclue_gaze_r50_l2cs_test.json')
    parser.add_argument('--anno', help='annotation json file', default='data/l2cs/test.json')
    parser.add_argument('--gaze-name', default='fusion_gazes', help='fusion_gazes | face_gazes | eyes_gazes | head_gazes')
    return parser.parse_args(argv)


def main(args):
    with open(args.evalfile) as f:
        eval_data = json.load(f)
    with open(args.anno) as f:
        anno_data = json.load(f)
    return metric.gaze_error(eval_data, anno_data, args.gaze_name, setting='l2cs')


if __name__ == '__main__':
    main(parse_args())
