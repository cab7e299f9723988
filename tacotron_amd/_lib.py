"""ctypes binding of libtaco_hip.so (include/taco.h). Nothing here computes: it marshals
pointers and sizes. The library must have been built (`python -m tacotron_amd.build`); there is
no CPU fallback - a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtaco_hip.so')

_HP_FIELDS = [
    'num_chars', 'embedded_depth', 'prenet_depth1', 'prenet_depth2',
    'encoder_K', 'encoder_bank_filters', 'encoder_proj1', 'encoder_proj2', 'encoder_proj_width',
    'encoder_highways', 'encoder_highway_depth', 'encoder_gru_cells',
    'attention_depth', 'decoder_depth', 'num_mels', 'outputs_per_step',
    'decoder_K', 'decoder_bank_filters', 'decoder_proj1', 'decoder_proj2', 'decoder_proj_width',
    'decoder_highways', 'decoder_highway_depth', 'decoder_gru_cells',
    'num_freq', 'n_fft', 'hop_length', 'win_length',
]


class TacoHParams(ctypes.Structure):
    _fields_ = ([('struct_size', ctypes.c_uint32)] + [(n, ctypes.c_int32) for n in _HP_FIELDS] +
                [('min_level_db', ctypes.c_float), ('ref_level_db', ctypes.c_float), ('power', ctypes.c_float),
                 ('max_iters', ctypes.c_int32), ('griffin_lim_iters', ctypes.c_int32),
                 ('encoder_pooling_width', ctypes.c_int32), ('encoder_pooling_stride', ctypes.c_int32),
                 ('decoder_pooling_width', ctypes.c_int32), ('decoder_pooling_stride', ctypes.c_int32)])


class TacoWeightEntry(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('data', ctypes.POINTER(ctypes.c_float)),
                ('ndim', ctypes.c_int32), ('dims', ctypes.c_int64 * 4)]


EXPORTS = ['taco_last_error', 'taco_create', 'taco_destroy', 'taco_workspace_bytes', 'taco_encode',
           'taco_decode', 'taco_decode_teacher', 'taco_postnet', 'taco_griffin_lim', 'taco_griffin_lim_noalloc', 'taco_griffin_lim_librosa', 'taco_spectrogram', 'taco_spectrogram_workspace_bytes', 'taco_synthesize', 'taco_timing_begin',
           'taco_timing_end', 'taco_kernel_name', 'taco_num_kernel_kinds', 'taco_set_decode_graph', 'taco_set_persist_budget', 'taco_get_persist_budget', 'taco_set_gl_split', 'taco_set_bigru_mode', 'taco_decode_status', 'taco_debug_conv_gemm', 'taco_debug_handoff_probe',
           'taco_debug_poison_decoder']

_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('%s is missing: build it with `python -m tacotron_amd.build` '
                           '(there is no CPU fallback)' % LIB_PATH)
    # torch first: it ships its own libamdhip64.so.7, and the stream / device pointers it hands us are
    # only meaningful to that copy of the HIP runtime. Loaded afterwards, libtaco_hip.so binds to it by soname.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
    lib.taco_last_error.restype = ctypes.c_char_p
    lib.taco_last_error.argtypes = []
    lib.taco_create.restype = ctypes.c_int
    lib.taco_create.argtypes = [ctypes.POINTER(TacoHParams), ctypes.POINTER(TacoWeightEntry), i32, i32,
                                ctypes.POINTER(vp)]
    lib.taco_destroy.restype = None
    lib.taco_destroy.argtypes = [vp]
    lib.taco_workspace_bytes.restype = sz
    lib.taco_workspace_bytes.argtypes = [vp, i32, i32, i32]
    lib.taco_encode.restype = ctypes.c_int
    lib.taco_encode.argtypes = [vp, vp, vp, i32, i32, vp, vp, sz, vp]
    lib.taco_decode.restype = ctypes.c_int
    lib.taco_decode.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]
    lib.taco_postnet.restype = ctypes.c_int
    lib.taco_postnet.argtypes = [vp, vp, i32, i32, vp, vp, sz, vp]
    lib.taco_griffin_lim.restype = ctypes.c_int
    lib.taco_griffin_lim.argtypes = [vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.taco_griffin_lim_noalloc.restype = ctypes.c_int
    lib.taco_griffin_lim_noalloc.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.taco_griffin_lim_librosa.restype = ctypes.c_int
    lib.taco_griffin_lim_librosa.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    lib.taco_synthesize.restype = ctypes.c_int
    lib.taco_synthesize.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.taco_timing_begin.restype = ctypes.c_int
    lib.taco_timing_begin.argtypes = [vp, i32]
    lib.taco_timing_end.restype = ctypes.c_int
    lib.taco_timing_end.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_double)]
    lib.taco_kernel_name.restype = ctypes.c_char_p
    lib.taco_kernel_name.argtypes = [i32]
    lib.taco_num_kernel_kinds.restype = i32
    lib.taco_num_kernel_kinds.argtypes = []
    lib.taco_set_decode_graph.restype = ctypes.c_int
    lib.taco_set_decode_graph.argtypes = [vp, i32]
    lib.taco_set_persist_budget.restype = ctypes.c_int
    lib.taco_set_persist_budget.argtypes = [i32, i32]
    lib.taco_set_gl_split.restype = ctypes.c_int
    lib.taco_set_gl_split.argtypes = [vp, i32]
    lib.taco_set_bigru_mode.restype = ctypes.c_int
    lib.taco_set_bigru_mode.argtypes = [vp, i32]
    lib.taco_get_persist_budget.restype = i32
    lib.taco_get_persist_budget.argtypes = [i32]
    lib.taco_spectrogram_workspace_bytes.restype = sz
    lib.taco_spectrogram_workspace_bytes.argtypes = [vp, i32, i32]
    lib.taco_spectrogram.restype = ctypes.c_int
    lib.taco_spectrogram.argtypes = [vp, vp, i32, i32, ctypes.c_float, vp, vp, vp, sz, vp]
    lib.taco_decode_teacher.restype = ctypes.c_int
    lib.taco_decode_teacher.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.taco_decode_status.restype = ctypes.c_int
    lib.taco_decode_status.argtypes = [vp]
    lib.taco_debug_poison_decoder.restype = ctypes.c_int
    lib.taco_debug_poison_decoder.argtypes = [vp]
    lib.taco_debug_handoff_probe.restype = ctypes.c_int
    lib.taco_debug_handoff_probe.argtypes = [i32, i32, ctypes.POINTER(ctypes.c_double)]
    lib.taco_debug_conv_gemm.restype = ctypes.c_int
    lib.taco_debug_conv_gemm.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp]
    _lib = lib
    return lib


def last_error():
    return load_library().taco_last_error().decode('utf-8', 'replace')


def make_hparams_struct(hp):
    from .text.characters import chars
    n_fft, hop, win = hp.stft_params()
    s = TacoHParams()
    s.struct_size = ctypes.sizeof(TacoHParams)
    s.num_chars = len(chars)
    s.embedded_depth = hp.embedded_depth
    s.prenet_depth1, s.prenet_depth2 = hp.prenet_depths
    s.encoder_K = hp.encoder_K
    s.encoder_bank_filters = hp.encoder_bank_num_filters
    s.encoder_proj1, s.encoder_proj2 = hp.encoder_proj_num_filters
    s.encoder_proj_width = hp.encoder_proj_filter_width
    s.encoder_highways = hp.encoder_num_highway_layers
    s.encoder_highway_depth = hp.encoder_highway_depth
    s.encoder_gru_cells = hp.encoder_gru_num_cells
    s.attention_depth = hp.attention_depth
    s.decoder_depth = hp.decoder_depth
    s.num_mels = hp.num_mels
    s.outputs_per_step = hp.outputs_per_step
    s.decoder_K = hp.decoder_K
    s.decoder_bank_filters = hp.decoder_bank_num_filters
    s.decoder_proj1, s.decoder_proj2 = hp.decoder_proj_num_filters
    s.decoder_proj_width = hp.decoder_proj_filter_width
    s.decoder_highways = hp.decoder_num_highway_layers
    s.decoder_highway_depth = hp.decoder_highway_depth
    s.decoder_gru_cells = hp.decoder_gru_num_cells
    s.num_freq = hp.num_freq
    s.n_fft, s.hop_length, s.win_length = n_fft, hop, win
    s.min_level_db = hp.min_level_db
    s.ref_level_db = hp.ref_level_db
    s.power = hp.power
    s.max_iters = hp.max_iters
    s.griffin_lim_iters = hp.griffin_lim_iters
    # carried so that taco_create can refuse values the kernels do not implement (modules.py:70-71)
    s.encoder_pooling_width, s.encoder_pooling_stride = hp.encoder_pooling_width, hp.encoder_pooling_stride
    s.decoder_pooling_width, s.decoder_pooling_stride = hp.decoder_pooling_width, hp.decoder_pooling_stride
    return s
