"""ctypes binding of ``libmcgaze_hip.so`` (C-ABI declared in ``include/mcgaze_hip.h``).

The product path has no CPU fallback: if the shared library is missing or fails to load,
``load()`` raises and every model built on it refuses to run.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MCGAZE_LIB') or os.path.join(_HERE, 'libmcgaze_hip.so')   # MCGAZE_LIB: A/B a second build on one box

MCG_OK = 0
MCG_F32, MCG_BF16, MCG_F16X3, MCG_F16 = 0, 1, 2, 3
ABI_VERSION = 13
RES_NONE, RES_ADD, RES_UPSAMPLE_ADD = 0, 1, 2
FLAG_STAGED_GEMM, FLAG_NO_SPECIALISED, FLAG_NO_ATTN_BLOCK = 1, 2, 4

# enum order of include/mcgaze_hip.h
STAGE_KEYS = [
    'IN_PROJ_W', 'IN_PROJ_B', 'OUT_PROJ_W', 'OUT_PROJ_B', 'ATTN_LN_G', 'ATTN_LN_B', 'DYN_W', 'DYN_B',
    'NORM_IN_G', 'NORM_IN_B', 'NORM_OUT_G', 'NORM_OUT_B', 'FC_W', 'FC_B', 'FC_LN_G', 'FC_LN_B', 'IIC_LN_G', 'IIC_LN_B',
    'FFN1_W', 'FFN1_B', 'FFN2_W', 'FFN2_B', 'FFN_LN_G', 'FFN_LN_B', 'CLS_FC_W', 'CLS_LN_G', 'CLS_LN_B',
    'REG_FC_W', 'REG_LN_G', 'REG_LN_B', 'HEAD_CLS_W', 'HEAD_CLS_B', 'HEAD_REG_W', 'HEAD_REG_B', 'OUT_PROJ_WF', 'CLS_FC_WF', 'REG_FC_WF', 'IN_PROJ_WF', 'DYN_WF']
GAZE_KEYS = ['FC_W', 'LN_G', 'LN_B', 'OUT_W', 'OUT_B', 'FUSE_W', 'FUSE_B']
SW_COUNT, GW_COUNT = len(STAGE_KEYS), len(GAZE_KEYS)

# every symbol include/mcgaze_hip.h declares
EXPORTS = ['mcg_abi_version', 'mcg_build_id', 'mcg_last_error', 'mcg_device_info', 'mcg_nchw_to_nhwc', 'mcg_nhwc_to_nchw', 'mcg_conv2d',
           'mcg_stem_workspace_bytes', 'mcg_stem_forward', 'mcg_roi_align', 'mcg_stage_workspace_bytes', 'mcg_stage_forward',
           'mcg_gaze_head_workspace_bytes', 'mcg_gaze_head', 'mcg_engine_create', 'mcg_engine_destroy',
           'mcg_engine_workspace_bytes', 'mcg_trunk_workspace_bytes', 'mcg_decoder_workspace_bytes', 'mcg_backbone_fpn_forward',
           'mcg_decoder_forward', 'mcg_clip_forward', 'mcg_preprocess_frames', 'mcg_engine_set_option', 'mcg_engine_profile_start',
           'mcg_engine_profile_stop', 'mcg_bench_backbone_forward', 'mcg_bottleneck_x3', 'mcg_bench_backbone_levels', 'mcg_conv3x3_wino_x3',
           'mcg_conv3x3_wino_x3_weight_bytes', 'mcg_engine_range_audit']


class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('bias', C.c_void_p), ('residual', C.c_void_p), ('y', C.c_void_p),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int), ('KH', C.c_int),
                ('KW', C.c_int), ('stride', C.c_int), ('pad', C.c_int), ('relu', C.c_int), ('residual_mode', C.c_int),
                ('Hr', C.c_int), ('Wr', C.c_int), ('x2', C.c_void_p), ('Cin2', C.c_int), ('stride2', C.c_int), ('H2', C.c_int),
                ('W2', C.c_int), ('tile', C.c_int), ('flags', C.c_int), ('wscale', C.c_float)]


class ConvWeights(C.Structure):
    _fields_ = [('w', C.c_void_p), ('bias', C.c_void_p), ('cin', C.c_int), ('cout', C.c_int), ('k', C.c_int),
                ('stride', C.c_int), ('pad', C.c_int), ('wf', C.c_void_p), ('wscale', C.c_float), ('wf4', C.c_void_p)]


class FusedBlock(C.Structure):
    _fields_ = [('wstream', C.c_void_p), ('bias', C.c_void_p), ('conv2_index', C.c_int), ('cm', C.c_int), ('c', C.c_int), ('cn', C.c_int),
                ('nsrc', C.c_int)]


class ModelWeights(C.Structure):
    _fields_ = [('blocks', C.c_int * 4), ('stem', ConvWeights), ('convs', C.POINTER(ConvWeights)), ('num_convs', C.c_int),
                ('lateral', ConvWeights * 4), ('fpn_out', ConvWeights * 4), ('c3_ds', ConvWeights * 4), ('init_boxes', C.c_void_p),
                ('init_feats', C.c_void_p), ('num_stages', C.c_int), ('stage_weights', C.POINTER(C.c_void_p)),
                ('gaze_weights', C.POINTER(C.c_void_p)), ('bbox_stds', C.c_float * 4), ('fused', C.POINTER(FusedBlock)), ('num_fused', C.c_int)]


class FrameDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('src_h', C.c_int), ('src_w', C.c_int), ('src_pitch', C.c_int), ('crop_y', C.c_int),
                ('crop_x', C.c_int), ('crop_h', C.c_int), ('crop_w', C.c_int), ('out_h', C.c_int), ('out_w', C.c_int)]


class McgError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises if it is absent: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McgError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       f'(or `make -C mcgaze_amd/csrc`); mcgaze_amd has no CPU fallback')
    lib = C.CDLL(LIB_PATH)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.mcg_abi_version.restype = i
    lib.mcg_last_error.restype = C.c_char_p
    lib.mcg_build_id.restype = C.c_char_p
    lib.mcg_device_info.argtypes = [C.POINTER(i), C.POINTER(sz), C.c_char_p, i]
    lib.mcg_nchw_to_nhwc.argtypes = [vp, i, vp, vp, i, i, i, i]
    lib.mcg_nhwc_to_nchw.argtypes = [vp, i, vp, vp, i, i, i, i]
    lib.mcg_conv2d.argtypes = [vp, i, C.POINTER(ConvDesc)]
    lib.mcg_stem_workspace_bytes.restype = sz
    lib.mcg_stem_workspace_bytes.argtypes = [i, i, i, i]
    lib.mcg_stem_forward.argtypes = [vp, i, vp, vp, vp, vp, i, i, i, vp, sz, i]
    lib.mcg_roi_align.argtypes = [vp, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i), C.POINTER(i), i, vp, i, i, vp, vp]
    lib.mcg_stage_workspace_bytes.restype = sz
    lib.mcg_stage_workspace_bytes.argtypes = [i, i]
    lib.mcg_stage_forward.argtypes = [vp, i, C.POINTER(vp), vp, vp, vp, i, i, vp, vp, vp, C.POINTER(C.c_float), vp, sz, i]
    lib.mcg_gaze_head_workspace_bytes.restype = sz
    lib.mcg_gaze_head_workspace_bytes.argtypes = [i, i]
    lib.mcg_gaze_head.argtypes = [vp, i, C.POINTER(vp), vp, i, vp, vp, sz]
    lib.mcg_engine_create.argtypes = [C.POINTER(vp), C.POINTER(ModelWeights), i]
    lib.mcg_engine_destroy.argtypes = [vp]
    lib.mcg_engine_destroy.restype = None
    lib.mcg_engine_workspace_bytes.restype = sz
    lib.mcg_engine_workspace_bytes.argtypes = [vp, i, i, i, i]
    lib.mcg_trunk_workspace_bytes.restype = sz
    lib.mcg_trunk_workspace_bytes.argtypes = [vp, i, i, i, i]
    lib.mcg_decoder_workspace_bytes.restype = sz
    lib.mcg_decoder_workspace_bytes.argtypes = [vp, i]
    lib.mcg_decoder_forward.argtypes = [vp, vp, C.POINTER(vp), i, i, i, i, vp, vp, vp, vp, vp, sz]
    lib.mcg_backbone_fpn_forward.argtypes = [vp, vp, vp, i, i, i, i, C.POINTER(vp), vp, sz]
    lib.mcg_clip_forward.argtypes = [vp, vp, vp, i, i, i, i, vp, i, vp, vp, vp, vp, sz]
    lib.mcg_preprocess_frames.argtypes = [vp, vp, i, vp, i, i, C.POINTER(C.c_float), C.POINTER(C.c_float), i]
    lib.mcg_engine_set_option.argtypes = [vp, C.c_char_p, i]
    lib.mcg_engine_profile_start.argtypes = [vp, i]
    lib.mcg_engine_profile_stop.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i), C.POINTER(i), i]
    lib.mcg_bench_backbone_forward.argtypes = [vp, vp, vp, i, i, i, vp, sz]
    lib.mcg_bench_backbone_levels.argtypes = [vp, vp, i, i, i, C.POINTER(vp)]
    lib.mcg_bottleneck_x3.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
    lib.mcg_conv3x3_wino_x3.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, C.c_float, i]
    lib.mcg_conv3x3_wino_x3_weight_bytes.restype = sz
    lib.mcg_engine_range_audit.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_char_p), i, C.POINTER(i)]
    lib.mcg_conv3x3_wino_x3_weight_bytes.argtypes = [i, i, i]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ('mcg_abi_version',):
            fn.restype = i
    if lib.mcg_abi_version() != ABI_VERSION:
        raise McgError(f'ABI mismatch: library reports {lib.mcg_abi_version()}, binding expects {ABI_VERSION}')
    _lib = lib
    return lib


def build_id():
    """The build id compiled into the loaded library (csrc/Makefile: sha256 over the kernel sources + the public header, 16 hex digits)."""
    return load().mcg_build_id().decode()


def source_id():
    """The same hash over the sources in THIS tree: differs from build_id() when the shared library is stale."""
    import hashlib
    src = os.path.join(_HERE, 'csrc')
    names = sorted(n for n in os.listdir(src) if n.endswith('.hip') or n.endswith('.hpp'))
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(src, n), 'rb').read())
    h.update(open(os.path.join(_HERE, '..', 'include', 'mcgaze_hip.h'), 'rb').read())
    return h.hexdigest()[:16]


def check(rc, what):
    if rc != MCG_OK:
        msg = load().mcg_last_error().decode(errors='replace')
        raise McgError(f'{what} failed (code {rc}): {msg}')
