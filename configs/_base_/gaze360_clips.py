# Gaze360 clip data settings (this repo's own config; same schema the reference's configs use).
# Frames are centre-cropped, resized so the long side is 224, normalised and padded to /32.
dataset_type = 'Gaze360Dataset'
data_root = 'data/gaze360/'
clip_length = 7

img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)

_frame_ops = [
    dict(type='CenterCrop', crop_size=(0.68, 0.68), crop_type='relative_range'),
    dict(type='Resize', img_scale=(224, 224), keep_ratio=True),
]
test_pipeline = [dict(type='LoadImageFromFile')] + _frame_ops + [
    dict(type='RandomFlip', flip_ratio=0.0),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img']),
]
data = dict(
    samples_per_gpu=64,
    workers_per_gpu=8,
    test=dict(type=dataset_type, ann_file=data_root + 'test.json', clip_length=clip_length,
              img_prefix=data_root + 'test_rawframes/', pipeline=test_pipeline))
