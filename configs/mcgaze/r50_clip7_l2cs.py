# MCGaze R-50, 7-frame clips, L2CS setting (this repo's own config, in the reference's schema): same model as the Gaze360
# setting; frames are NOT cropped, resized so that the long side is 448 (aspect kept -> non-square frames), padded to /32.
_base_ = './r50_clip7_gaze360.py'

clip_length = 7
dataset_type = 'Gaze360Dataset'
data_root = 'data/l2cs/'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)

test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='Resize', img_scale=(448, 448), keep_ratio=True),
    dict(type='RandomFlip', flip_ratio=0.0),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img']),
]
data = dict(
    samples_per_gpu=16,
    test=dict(_delete_=True, type=dataset_type, ann_file=data_root + 'test.json', clip_length=clip_length,
              img_prefix=data_root + 'test_rawframes/', pipeline=test_pipeline))
