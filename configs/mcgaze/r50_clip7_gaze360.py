# MCGaze R-50, 7-frame clips (this repo's own config, written to the reference's schema so that it
# and the reference's configs/multiclue_gaze/*.py are interchangeable).
_base_ = ['../_base_/gaze360_clips.py', '../_base_/runtime.py']

num_stages = 4
clip_length = 7
_d = 256


def _stage():
    return dict(
        type='GazeSTQIHead', num_classes=3, in_channels=_d, num_heads=8, num_ffn_fcs=2, feedforward_channels=2048,
        num_cls_fcs=1, num_reg_fcs=3, dropout=0.0, ffn_act_cfg=dict(type='ReLU', inplace=True),
        dynamic_conv_cfg=dict(type='DynamicConv', in_channels=_d, feat_channels=64, out_channels=_d, input_feat_shape=7,
                              act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN')),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_bbox=dict(type='L1Loss', loss_weight=5.0), loss_iou=dict(type='GIoULoss', loss_weight=2.0),
        bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=False, target_means=[0., 0., 0., 0.], target_stds=[0.5, 0.5, 1., 1.]))


def _gaze():
    return dict(type='GazeHead', in_channels=_d, loss_gaze=dict(type='GazeArccosLoss', loss_weight=6.0),
                loss_temp=dict(type='GazeTempLoss', clip_len=clip_length, loss_weight=1.0))


model = dict(
    type='MultiClueGaze',
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch', init_cfg=None),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=_d, start_level=0, add_extra_convs='on_input', num_outs=4),
    rpn_head=dict(type='FixedEmbeddingRPNHead', proposal_feature_channel=_d),
    roi_head=dict(
        type='MultiClueGazeROIHead', num_stages=num_stages, stage_loss_weights=[1] * num_stages, proposal_feature_channel=_d,
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2),
                                out_channels=_d, featmap_strides=[4, 8, 16, 32]),
        bbox_head=[_stage() for _ in range(num_stages)],
        gaze_head=[_gaze() for _ in range(num_stages)]),
    train_cfg=None,
    test_cfg=dict(rpn=None, rcnn=dict(max_per_img=2, mask_thr_binary=0.5)))
