"""ctypes binding of libpeahip.so (C ABI: include/peahip.h).

The product path has NO fallback: if the HIP library is missing or no gfx950 device is visible, every
compute entry point raises.  (CPU parity checking lives in oracle/, which this package never imports.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PEA_LIB') or os.path.join(_HERE, 'csrc', 'libpeahip.so')   # PEA_LIB: A/B runs of two builds

PEA_OK = 0
KIND_GAT, KIND_GCN, KIND_SAGE = 0, 1, 2
PLAN_SELF_LOOPS = 1
PLAN_EDGE_IDS = 2
FUSE_ATT, FUSE_MEAN = 0, 1

_ERR_NAMES = {-1: 'bad argument', -2: 'id out of range', -3: 'HIP runtime error', -4: 'workspace too small',
              -5: 'no gfx950 device'}


class PeaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('peahip error %d (%s): %s' % (code, _ERR_NAMES.get(code, '?'), msg))
        self.code = code


class ModelDesc(C.Structure):
    _fields_ = [('kind', C.c_int), ('num_channels', C.c_int), ('steps', C.POINTER(C.c_int)),
                ('relation_of', C.POINTER(C.c_int)), ('emb_dim', C.c_int), ('hidden_size', C.c_int),
                ('repr_dim', C.c_int), ('heads', C.c_int), ('fuse_mode', C.c_int),
                ('gcn_deg_from_col', C.c_int), ('negative_slope', C.c_float), ('enable_backward', C.c_int),
                ('reverse_of', C.POINTER(C.c_int))]


class ExchangeDesc(C.Structure):
    _fields_ = [('relation', C.c_int), ('slots_per_rank', C.c_int64), ('width', C.c_int), ('dst_ld', C.c_int),
                ('dst_offset_bytes', C.c_size_t), ('src_offset_bytes', C.c_size_t), ('src_ld', C.c_int),
                ('src_col', C.c_int)]


class GwJob(C.Structure):
    _fields_ = [('a', C.c_void_p), ('lda', C.c_int64), ('ma', C.c_int), ('b', C.c_void_p), ('ldb', C.c_int64),
                ('nb', C.c_int), ('out', C.c_void_p), ('ldo', C.c_int64), ('b_mask', C.c_void_p), ('b_alt', C.c_void_p),
                ('ldb_alt', C.c_int64), ('b_alt_scale', C.c_void_p)]


class XchgJob(C.Structure):
    _fields_ = [('table', C.c_void_p), ('ld', C.c_int64), ('col', C.c_int), ('width', C.c_int), ('nodes', C.c_void_p),
                ('slots', C.c_void_p), ('n', C.c_int64), ('buf_off', C.c_int64), ('slots_per_rank', C.c_int)]


class DenseJob(C.Structure):
    _fields_ = [('a', C.c_void_p), ('lda', C.c_int64), ('k', C.c_int), ('w', C.c_void_p), ('ldw', C.c_int64),
                ('n_out', C.c_int), ('out', C.c_void_p), ('ldo', C.c_int64), ('gate', C.c_void_p), ('ld_gate', C.c_int64)]


class Mlp2BwdChan(C.Structure):
    _fields_ = [('w0', C.c_void_p), ('w1', C.c_void_p), ('dt1_col', C.c_int), ('h_col', C.c_int), ('dz_col', C.c_int),
                ('da_col', C.c_int)]


class Mlp2BwdChanSage(C.Structure):
    _fields_ = [('w0_root', C.c_void_p), ('w1_root', C.c_void_p), ('dr1_col', C.c_int), ('dxr_col', C.c_int)]


class StageOpts(C.Structure):
    _fields_ = [('part', C.c_int), ('sel_ids', C.c_void_p), ('sel_stride', C.c_int64), ('n_sel', C.c_int64),
                ('sel_out', C.c_void_p), ('err_flag', C.c_void_p)]


PART_ALL, PART_SOURCES, PART_REST = 0, 1, 2     # include/peahip.h PEA_PART_*
BWD_PREMASKED = 0x100     # include/peahip.h PEA_BWD_PREMASKED

# every symbol include/peahip.h declares: name -> (restype, argtypes)
_vp, _i64, _int, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
SIGNATURES = {
    'pea_version': (C.c_char_p, []),
    'pea_last_error': (C.c_char_p, []),
    'pea_device_count': (_int, []),
    'pea_plan_create': (_int, [_i64, _int, C.POINTER(_vp), C.POINTER(_i64), _int, _int, _int, _int, _int, _vp, C.POINTER(_vp)]),
    'pea_plan_destroy': (_int, [_vp]),
    'pea_plan_relation_info': (_int, [_vp, _int, C.POINTER(_i64)]),
    'pea_plan_export_csr': (_int, [_vp, _int, _vp, _vp, _vp]),
    'pea_model_create': (_int, [_vp, C.POINTER(ModelDesc), C.POINTER(_vp)]),
    'pea_model_destroy': (_int, [_vp]),
    'pea_model_workspace_bytes': (_sz, [_vp]),
    'pea_model_params_per_layer': (_int, [_vp]),
    'pea_model_forward': (_int, [_vp, C.POINTER(_vp), _vp, _vp, _int, _vp, _sz, _vp, _vp, _vp]),
    'pea_plan_set_owned_rows': (_int, [_vp, _vp, _i64, _vp]),
    'pea_plan_set_sources': (_int, [_vp, _int, _vp, _i64, _vp, _i64, _vp]),
    'pea_model_num_stages': (_int, [_vp]),
    'pea_model_forward_stage': (_int, [_vp, _int, C.POINTER(_vp), _vp, _vp, _int, _vp, _sz, _vp, _vp, _vp]),
    'pea_model_forward_stage_train': (_int, [_vp, _int, C.POINTER(_vp), _vp, _vp, _int, _vp, _sz, _vp, _vp, _vp]),
    'pea_plan_set_owned_split': (_int, [_vp, _i64]),
    'pea_model_forward_part': (_int, [_vp, _int, C.POINTER(StageOpts), C.POINTER(_vp), _vp, _vp, _int, _vp, _sz, _vp, _vp, _vp]),
    'pea_model_stage_fills_exchange': (_int, [_vp, _int]),
    'pea_tape_create': (_int, [C.POINTER(_vp)]),
    'pea_tape_destroy': (_int, [_vp]),
    'pea_tape_begin': (_int, [_vp]),
    'pea_tape_end': (_int, [_vp]),
    'pea_tape_length': (_int, [_vp]),
    'pea_tape_replay': (_int, [_vp, _vp]),
    'pea_model_num_exchanges': (_int, [_vp, _int]),
    'pea_model_exchange_desc': (_int, [_vp, _int, _int, C.POINTER(ExchangeDesc)]),
    'pea_model_forward_train': (_int, [_vp, C.POINTER(_vp), _vp, _vp, _int, _vp, _sz, _vp, _vp, _vp]),
    'pea_model_backward_level': (_int, [_vp, _int, _int, _vp, _sz, _vp]),
    'pea_model_set_active_rows': (_int, [_vp, _vp]),
    'pea_model_describe': (_int, [_vp, C.POINTER(_i64), _int, C.POINTER(_int)]),
    'pea_model_stats': (_int, [_vp, C.POINTER(_i64), C.POINTER(C.c_double)]),
    'pea_model_compulsory_bytes': (C.c_double, [_vp]),
    'pea_conv_workspace_bytes': (_sz, [_vp, _int, _int, _int, _int, _int]),
    'pea_gat_conv': (_int, [_vp, _int, _int, _int, _int, _vp, _i64, _vp, _vp, _vp, _vp, C.c_float, _int, _vp, _i64, _vp, _sz, _vp]),
    'pea_gcn_conv': (_int, [_vp, _int, _int, _int, _vp, _i64, _vp, _vp, _int, _int, _vp, _i64, _vp, _sz, _vp]),
    'pea_sage_conv': (_int, [_vp, _int, _int, _int, _vp, _i64, _vp, _vp, _vp, _int, _vp, _i64, _vp, _sz, _vp]),
    'pea_weighted_aggregate_workspace_bytes': (_sz, [_vp, _int, _int]),
    'pea_weighted_aggregate': (_int, [_vp, _int, _int, _vp, _i64, _vp, _vp, _i64, _vp, _sz, _vp]),
    'pea_grad_weight_workspace_bytes': (_sz, []),
    'pea_grad_weight': (_int, [_i64, _int, C.POINTER(GwJob), _vp, _sz, _vp]),
    'pea_dense_batch': (_int, [_i64, _int, C.POINTER(DenseJob), _vp]),
    'pea_mlp2_backward_data_workspace_bytes': (_sz, [_int, _int, _int, _int]),
    'pea_mlp2_backward_data': (_int, [_i64, _int, C.POINTER(Mlp2BwdChan), _int, _int, _int, _vp, _i64, _vp, _i64, _vp, _i64, _vp,
                                      _i64, _vp, _vp, _int, _vp, _sz, _vp]),
    'pea_rows_nonzero_workspace_bytes': (_sz, [_i64]),
    'pea_rows_nonzero': (_int, [_i64, _int, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pea_rows_nonzero_or': (_int, [_i64, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pea_mlp2_backward_data_sage': (_int, [_i64, _int, C.POINTER(Mlp2BwdChan), C.POINTER(Mlp2BwdChanSage), _int, _int, _int,
                                           _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _sz,
                                           _vp]),
    'pea_grad_weight_rows': (_int, [_i64, _vp, _vp, _i64, _int, C.POINTER(GwJob), _vp, _sz, _vp]),
    'pea_model_set_active_rows0': (_int, [_vp, _vp, _vp, _vp]),
    'pea_rows_zero': (_int, [_vp, _i64, _int, _vp, _vp, _vp]),
    'pea_block_sum': (_int, [_i64, _int, _int, _vp, _i64, _vp, _i64, _vp]),
    'pea_grad_weight_sharded': (_int, [_i64, _int, _int, _int, _int, C.POINTER(GwJob), _vp, _sz, _vp]),
    'pea_dense_batch_rows': (_int, [_i64, _vp, _int, C.POINTER(DenseJob), _vp]),
    'pea_sample_negatives': (_int, [_i64, _int, _vp, _vp, _i64, _i64, _vp, _i64, C.c_uint64, C.c_uint32, _vp, _i64, _vp, _vp]),
    'pea_fuse': (_int, [_i64, _int, _int, _vp, _i64, C.POINTER(_int), _vp, _int, _int, _vp, _vp]),
    'pea_bpr_workspace_bytes': (_sz, [_i64]),
    'pea_bpr_score': (_int, [_i64, _int, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pea_predict': (_int, [_i64, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pea_profile_enable': (_int, [_int]),
    'pea_profile_read': (_int, [_int, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(_int)]),
    'pea_profile_read_ex': (_int, [_int, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), C.POINTER(_int)]),
    'pea_entity_reg_workspace_bytes': (C.c_size_t, [_i64, _int]),
    'pea_entity_reg': (_int, [_i64, _int, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, C.c_size_t, _vp]),
    'pea_bpr_train_workspace_bytes': (C.c_size_t, [_i64]),
    'pea_bpr_train_supported': (_int, [_int, _int]),
    'pea_bpr_train': (_int, [_i64, _int, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    'pea_rows_scatter_sum_workspace_bytes': (C.c_size_t, [_i64]),
    'pea_rows_scatter_sum': (_int, [_i64, _vp, _vp, _i64, _int, _int, C.POINTER(_int), _vp, _i64, _i64, _vp, C.c_size_t, _vp]),
    'pea_rows_pack': (_int, [_vp, _i64, _int, _int, _vp, _i64, _vp, _i64, _vp]),
    'pea_rows_unpack': (_int, [_vp, _i64, _vp, _int, _vp, _i64, _vp, _i64, _int, _vp]),
    'pea_rows_pack_batch': (_int, [_int, C.POINTER(XchgJob), _vp, _vp]),
    'pea_rows_unpack_batch': (_int, [_int, C.POINTER(XchgJob), _vp, _i64, _vp]),
    'pea_rows_select_owned': (_int, [_vp, _i64, _int, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _vp, _vp]),
    'pea_rank_eval': (_int, [_i64, _int, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Loads the shared library and types every entry point.  Raises if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(or make -C graph_recsys_benchmark_amd/csrc); there is no CPU fallback' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().pea_last_error().decode()


def check(rc):
    if rc != PEA_OK:
        raise PeaError(rc, last_error())


def require_device():
    lib = load()
    if lib.pea_device_count() <= 0:
        raise PeaError(-5, 'no gfx950 (MI355X) device visible; the HIP path has no CPU fallback')
    return lib


class Tape:
    """A recorded launch sequence (include/peahip.h, pea_tape_*): `with tape.record(): <library calls>` once, then
    tape.replay() while the pointers those calls were given stay valid."""

    def __init__(self):
        self._h = C.c_void_p()
        check(load().pea_tape_create(C.byref(self._h)))

    def record(self):
        return _TapeRecording(self)

    def replay(self):
        check(load().pea_tape_replay(self._h, current_stream()))

    def __len__(self):
        return int(load().pea_tape_length(self._h))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                load().pea_tape_destroy(h)
            except Exception:
                pass


class _TapeRecording:
    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        check(load().pea_tape_begin(self.tape._h))
        return self.tape

    def __exit__(self, *exc):
        load().pea_tape_end(self.tape._h)
        return False


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = None      # torch's raw current-stream getter (one C call; torch.cuda.current_stream() builds a Stream object
                        # and resolves the device through several Python layers: ~3 us, six times per step)


def current_stream():
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
