"""Gaze360 mean angular error of a result file -- the reference's command line (tools/calculate_mae_gaze360.py:7-14,96-105:
``--evalfile results/results_<cfg>_<json> [--anno data/gaze360/test.json]``, prints the three ``fusion_gazes`` lines of
:185-187).  The arithmetic is mcgaze_amd.metric.gaze_error."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcgaze_amd import metric  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Gaze360 MAE of a result json')
    parser.add_argument('--evalfile', help='pred_gaze json file', default='results/results_multiclue_gaze_r50_gaze360_test.json')
    parser.add_argument('--anno', help='annotation json file', default='data/gaze360/test.json')
    parser.add_argument('--gaze-name', default='fusion_gazes', help='fusion_gazes | face_gazes | eyes_gazes | head_gazes')
    return parser.parse_args(argv)


def main(args):
    with open(args.evalfile) as f:
        eval_data = json.load(f)
    with open(args.anno) as f:
        anno_data = json.load(f)
    return metric.gaze_error(eval_data, anno_data, args.gaze_name, setting='gaze360')


if __name__ == '__main__':
    main(parse_args())
