"""ctypes binding of libmmt_hip.so (the C ABI declared in include/mmt_hip.h).

There is NO fallback: if the library is missing the import of any product module fails loudly, and
calling a kernel without a GPU raises.  The oracle is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MMT_HIP_LIB') or os.path.join(_HERE, 'lib', 'libmmt_hip.so')  # env override: lab builds only

c_int, c_i64, c_u32, c_f32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p


class MmtEpilogue(ctypes.Structure):
  _fields_ = [('bias', c_vp), ('res', c_vp), ('ldres', c_i64), ('out2', c_vp), ('ldout2', c_i64),
              ('aux', c_vp), ('ldaux', c_i64), ('colsum', c_vp), ('row_index', c_vp), ('seed_dev', c_vp),
              ('drop_key', c_u32), ('drop_thr16', c_u32), ('drop_scale', c_f32), ('reserved', ctypes.c_int32),
              ('dot_src', c_vp), ('lddot', c_i64), ('dot_out', c_vp),
              ('rider', c_vp), ('rider_limit', ctypes.c_int32), ('rider_slot', ctypes.c_int32),
              ('live_rows_hint', ctypes.c_int32), ('rider_cap', ctypes.c_int32)]


class MmtPackItem(ctypes.Structure):
  _fields_ = [('src', c_vp), ('dst', c_vp), ('dst_t', c_vp), ('rows', ctypes.c_int32), ('cols', ctypes.c_int32),
              ('dst_ld', ctypes.c_int32), ('dst_t_ld', ctypes.c_int32), ('dst_t_rows', ctypes.c_int32),
              ('reserved', ctypes.c_int32)]


class MmtAdamSeg(ctypes.Structure):
  _fields_ = [('offset', c_i64), ('count', c_i64), ('dst', c_vp), ('dst_t', c_vp), ('rows', ctypes.c_int32),
              ('cols', ctypes.c_int32), ('dst_ld', ctypes.c_int32), ('dst_t_ld', ctypes.c_int32)]


RIDER_SLOTS, RIDER_STAGES = 1024, 72
RIDER_STAT0 = RIDER_STAGES + 1 + RIDER_SLOTS              # first statistics word of MmtAdamQueue.state
RIDER_STATE_WORDS = RIDER_STAT0 + 64 + 64 * 64


class MmtAdamQueue(ctypes.Structure):
  _fields_ = [('p', c_vp), ('m', c_vp), ('v', c_vp), ('g', c_vp), ('segs', c_vp), ('unit_seg', c_vp), ('unit_blk', c_vp),
              ('state', c_vp), ('step_dev', c_vp), ('lr_dev', c_vp), ('chain', c_vp),
              ('lr', c_f32), ('beta1', c_f32), ('beta2', c_f32), ('eps', c_f32), ('weight_decay', c_f32),
              ('n_units', ctypes.c_int32), ('chain_stages', ctypes.c_int32), ('n_stages', ctypes.c_int32),
              ('stage_begin', ctypes.c_int32 * (RIDER_STAGES + 1))]


class MmtGemmItem(ctypes.Structure):
  _fields_ = [('A', c_vp), ('B', c_vp), ('C', c_vp), ('bias', c_vp), ('lda', c_i64), ('ldb', c_i64), ('ldc', c_i64),
              ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('tile_begin', ctypes.c_int32),
              ('n_rows_dev', c_vp)]


class MmtWgradItem(ctypes.Structure):
  _fields_ = [('A', c_vp), ('B', c_vp), ('out', c_vp), ('bias_out', c_vp), ('lda', c_i64), ('ldb', c_i64), ('ldo', c_i64),
              ('N', ctypes.c_int32), ('K2', ctypes.c_int32), ('N_out', ctypes.c_int32), ('K2_out', ctypes.c_int32),
              ('tile_begin', ctypes.c_int32), ('reserved', ctypes.c_int32), ('slab', c_vp), ('bias_slab', c_vp),
              ('splits', ctypes.c_int32), ('reserved2', ctypes.c_int32), ('n_rows_dev', c_vp)]


class MmtWgradGroup(ctypes.Structure):
  _fields_ = [('item', MmtWgradItem * 16), ('n_rows_dev', c_vp), ('count', ctypes.c_int32), ('rows', ctypes.c_int32)]


class MmtColReduceJob(ctypes.Structure):
  _fields_ = [('partials', c_vp), ('out', c_vp * 4), ('nblocks', ctypes.c_int32), ('nvec', ctypes.c_int32),
              ('nout', ctypes.c_int32), ('d', ctypes.c_int32)]


class MmtVideoSrc(ctypes.Structure):
  _fields_ = [('src_row', c_vp), ('src_cnt', c_vp), ('xsrc', c_vp)]


class MmtExpertIO(ctypes.Structure):
  _fields_ = [('feat', c_vp), ('maxpool', c_vp), ('ind', c_vp), ('t', c_vp), ('x', c_vp), ('y', c_vp), ('dy', c_vp),
              ('D', ctypes.c_int32), ('Dpad', ctypes.c_int32), ('type_idx', ctypes.c_int32),
              ('rows_pad', ctypes.c_int32), ('y_part', c_vp * 2), ('n_part', ctypes.c_int32),
              ('reserved', ctypes.c_int32)]


_LAYER_PTRS = ['wqkv', 'wqkv_t', 'wo', 'wo_t', 'w1', 'w1_t', 'w2', 'w2_t',
               'bqkv', 'bo', 'ln1_g', 'ln1_b', 'b1', 'b2', 'ln2_g', 'ln2_b',
               'g_wqkv', 'g_bqkv', 'g_wo', 'g_bo', 'g_ln1_g', 'g_ln1_b', 'g_w1', 'g_b1', 'g_w2', 'g_b2',
               'g_ln2_g', 'g_ln2_b']


class MmtBertLayer(ctypes.Structure):
  _fields_ = [(n, c_vp) for n in _LAYER_PTRS]


class MmtBertModel(ctypes.Structure):
  _fields_ = ([(n, ctypes.c_int32) for n in ('hidden', 'layers', 'heads', 'inter', 'max_pos', 'type_vocab')] +
              [(n, c_f32) for n in ('ln_eps', 'p_hidden', 'p_attn')] + [('reserved', ctypes.c_int32)] +
              [(n, c_vp) for n in ('pos_emb', 'type_emb', 'emb_ln_g', 'emb_ln_b', 'g_pos_emb', 'g_type_emb',
                                   'g_emb_ln_g', 'g_emb_ln_b')] +
              [('layer', ctypes.POINTER(MmtBertLayer))])


class MmtBertBatch(ctypes.Structure):
  _fields_ = ([(n, c_vp) for n in ('features', 'type_ids', 'pos_ids', 'mask_bias', 'cu_seqlens', 'row_index',
                                   'n_rows_dev', 'seed_dev')] +
              [(n, ctypes.c_int32) for n in ('rows', 'rows_alloc', 'batch', 'seq')] +
              [('out_rows', c_vp), ('n_out_per_sample', ctypes.c_int32), ('fork', ctypes.c_int32),
               ('side_stream', c_vp), ('rider', c_vp), ('rider_limits', c_vp), ('rider_slot0', ctypes.c_int32),
               ('live_rows_hint', ctypes.c_int32)])


FORK_WGRAD, FORK_EARLY, FORK_REDUCE, FORK_JOIN, RANGE_LAYERS_ONLY = 1, 2, 4, 8, 16


_PTR16 = c_vp * 16


class MmtSgemm(ctypes.Structure):
  _fields_ = [('A', _PTR16), ('B', _PTR16), ('C', _PTR16), ('bias', _PTR16),
              ('sai', c_i64), ('sak', c_i64), ('sbj', c_i64), ('sbk', c_i64), ('ldc', c_i64),
              ('batch', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
              ('beta', c_f32), ('reserved', ctypes.c_int32)]


_TEXT_HEAD_FIELDS = ['w1', 'b1', 'w2', 'b2', 'bn_gamma', 'bn_beta', 'running_mean', 'running_var', 'moe_w', 'moe_b',
                     'g_w1', 'g_b1', 'g_w2', 'g_b2', 'g_bn_gamma', 'g_bn_beta', 'g_moe_w', 'g_moe_b']


class MmtTextHeads(ctypes.Structure):
  _fields_ = [(n, _PTR16) for n in _TEXT_HEAD_FIELDS]


class MmtVideoFront(ctypes.Structure):
  _fields_ = [('experts', c_vp), ('M', ctypes.c_int32), ('B', ctypes.c_int32), ('T', ctypes.c_int32), ('pack', ctypes.c_int32),
              ('max_pos', ctypes.c_int32), ('do_cast', ctypes.c_int32), ('counts', c_vp), ('cu_seqlens', c_vp),
              ('n_rows_dev', c_vp), ('slot', c_vp), ('row_index', c_vp), ('type_ids', c_vp), ('pos_ids', c_vp),
              ('mask_bias', c_vp), ('agg_row', c_vp), ('seed_bump', c_vp), ('src', c_vp)]


class MmtTextHeadsOpts(ctypes.Structure):
  _fields_ = [('moe_drop_key', c_u32), ('moe_drop_thr16', c_u32), ('moe_drop_scale', c_f32), ('reserved', ctypes.c_int32),
              ('seed_dev', c_vp), ('key_dev', c_vp), ('num_batches_tracked', c_vp), ('video_front', c_vp)]


EPI = dict(BF16=0, BIAS_BF16=1, BIAS_GELU=2, BIAS_DROP_RES=3, DGELU=4, ADD_F32=5, F32=6, BIAS_F32=7)

# name -> (restype, argtypes); must list every symbol declared in include/mmt_hip.h
SIGNATURES = {
    'mmt_abi_version': (c_int, []),
    'mmt_build_info': (ctypes.c_char_p, []),
    'mmt_gemm_nt_bf16': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int,
                                 ctypes.POINTER(MmtEpilogue), c_vp, c_vp]),
    'mmt_gemm_select_tile': (c_int, [c_int] * 9),
    'mmt_gemm_tn_bf16': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_gemm_nt_splitk_workspace_floats': (c_i64, [c_int, c_int, c_int]),
    'mmt_gemm_nt_splitk': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int,
                                   ctypes.POINTER(MmtEpilogue), c_vp, c_vp]),
    'mmt_gemm_nn_bf16': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int,
                                 ctypes.POINTER(MmtEpilogue), c_vp, c_vp]),
    'mmt_gemm_nn_splitk_ex': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int,
                                      ctypes.POINTER(MmtEpilogue), c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    'mmt_gemm_nt_splitk_ex': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int,
                                      ctypes.POINTER(MmtEpilogue), c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    'mmt_gemm_nt_grouped': (c_int, [ctypes.POINTER(MmtGemmItem), c_int, c_int, c_vp]),
    'mmt_wgrad_grouped': (c_int, [ctypes.POINTER(MmtWgradGroup), c_vp]),
    'mmt_reduce_slabs': (c_int, [c_vp, c_int, c_i64, c_vp, c_int, c_vp]),
    'mmt_reduce_slabs_pair': (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_int, c_vp]),
    'mmt_ln_fwd': (c_int, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    'mmt_gemm_nt_ln_fwd': (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp,
                                   c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_embed_ln_fwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                                 c_int, c_int, c_vp, c_vp, c_u32, c_u32, c_f32, c_vp, c_vp]),
    'mmt_embed_ln_fwd_sched': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                                 c_int, c_int, c_vp, c_vp, c_u32, c_u32, c_f32, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_ln_bwd_rows_per_block': (c_int, [c_int]),
    'mmt_gemm_splitk_geometry': (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_splitk_ln_fwd': (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp,
                                  c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    'mmt_ln_bwd_slabs': (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp,
                                 c_u32, c_u32, c_f32, c_vp, c_vp]),
    'mmt_splitk_ln_fwd_ex': (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp,
                                     c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    'mmt_ln_bwd_slabs_ex': (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp,
                                    c_vp, c_u32, c_u32, c_f32, c_vp, c_vp]),
    'mmt_ln_bwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp,
                           c_u32, c_u32, c_f32, c_vp, c_vp]),
    'mmt_col_reduce': (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    'mmt_col_reduce_multi': (c_int, [ctypes.POINTER(MmtColReduceJob), c_int, c_vp]),
    'mmt_table_grad_partials': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_table_grad_chunks': (c_int, []),
    'mmt_table_grad_partials_pair': (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    'mmt_table_grad_scratch_floats': (c_i64, [c_int, c_int]),
    'mmt_table_grad': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    'mmt_attn_fwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_u32, c_u32,
                             c_f32, c_vp, c_vp, c_vp]),
    'mmt_attn_bwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32,
                             c_u32, c_u32, c_f32, c_vp, c_vp, c_vp]),
    'mmt_attn_bwd_ex': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_f32,
                                c_u32, c_u32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    'mmt_attn_schedule_words': (c_i64, [c_int, c_int, c_int]),
    'mmt_attn_schedule': (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_attn_bwd_rows_ex': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                     c_int, c_f32, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp]),
    'mmt_attn_fwd_rows': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_u32, c_u32,
                                  c_f32, c_vp, c_vp, c_vp]),
    'mmt_attn_bwd_rows': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                  c_f32, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp]),
    'mmt_ln_fwd_scatter': (c_int, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    'mmt_rows_gather': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    'mmt_rows_scatter': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    'mmt_embedding_grad': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_text_plan': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mmt_table_grad_direct': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_attn_dropout_mask': (c_int, [c_vp, c_int, c_int, c_int, c_u32, c_u32, c_vp, c_vp]),
    'mmt_reduce_slabs_2d': (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    'mmt_colsum_bf16': (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_pack_weights': (c_int, [ctypes.POINTER(MmtPackItem), c_int, c_vp]),
    'mmt_dropout_f32': (c_int, [c_vp, c_vp, c_i64, c_u32, c_u32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    'mmt_debug_dispatch_probe': (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_adam_step': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp]),
    'mmt_adam_fused_blocks': (c_int, [ctypes.POINTER(MmtAdamSeg)]),
    'mmt_adam_step_fused': (c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(MmtAdamSeg), c_vp, c_int, c_f32, c_f32, c_f32,
                                    c_f32, c_f32, c_vp, c_vp, c_int, c_vp]),
    'mmt_adam_step_queue': (c_int, [ctypes.POINTER(MmtAdamQueue), c_vp, c_vp]),
    'mmt_adam_rider_probe': (c_int, [c_vp, c_int, c_int, c_vp]),
    'mmt_video_plan': (c_int, [ctypes.POINTER(MmtExpertIO), c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                               c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(MmtVideoSrc), c_vp]),
    'mmt_video_cast': (c_int, [ctypes.POINTER(MmtExpertIO), c_int, c_int, c_int, ctypes.POINTER(MmtVideoSrc), c_vp]),
    'mmt_video_scatter': (c_int, [ctypes.POINTER(MmtExpertIO), c_int, c_int, c_int, c_int, c_vp, c_vp,
                                  ctypes.POINTER(MmtVideoSrc), c_vp, c_vp]),
    'mmt_video_scatter_bwd': (c_int, [ctypes.POINTER(MmtExpertIO), c_int, c_int, c_int, c_int, c_vp, c_vp,
                                      ctypes.POINTER(MmtVideoSrc), c_vp, c_vp]),
    'mmt_readout_fwd': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_readout_bwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    'mmt_sims_fwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_sims_bwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                             c_vp]),
    'mmt_sims_eval_workspace_floats': (c_i64, [c_int, c_int, c_int, c_int]),
    'mmt_sims_eval': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_retrieval_ranks': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    'mmt_ls_fold_bf16': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_transpose_bf16': (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_i64, c_vp]),
    'mmt_ls_finish': (c_int, [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'mmt_ls_col_blocks': (c_int, [c_int]),
    'mmt_ls_diag': (c_int, [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mmt_ls_counts_ex': (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_vp]),
    'mmt_ls_counts': (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_vp]),
    'mmt_ls_grad': (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_f32, c_vp, c_i64,
                            c_vp, c_vp]),
    'mmt_ls_grad_ex': (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_f32, c_vp, c_i64,
                            c_vp, c_int, c_vp]),
    'mmt_ls_unfold': (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    'mmt_simloss_small_max_n': (c_int, []),
    'mmt_simloss_bwd_small': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_int, c_vp, c_vp,
                                      c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mmt_maxmargin': (c_int, [c_vp, c_int, c_f32, c_int, c_vp, c_vp, c_vp, c_vp]),
    'mmt_infonce': (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    'mmt_bert_tail_capacity': (c_int, [c_int]),
    'mmt_bert_workspace_bytes': (c_i64, [ctypes.POINTER(MmtBertModel), c_int]),
    'mmt_bert_forward': (c_int, [ctypes.POINTER(MmtBertModel), ctypes.POINTER(MmtBertBatch), c_vp, c_vp, c_int, c_vp]),
    'mmt_bert_backward': (c_int, [ctypes.POINTER(MmtBertModel), ctypes.POINTER(MmtBertBatch), c_vp, c_vp, c_vp, c_int,
                                  c_vp]),
    'mmt_bert_backward_range': (c_int, [ctypes.POINTER(MmtBertModel), ctypes.POINTER(MmtBertBatch), c_vp, c_vp, c_vp,
                                        c_int, c_int, c_int, c_vp]),
    'mmt_stream_fork': (c_int, [c_vp, c_vp]),
    'mmt_sgemm_batched': (c_int, [ctypes.POINTER(MmtSgemm), c_vp]),
    'mmt_text_heads_workspace_floats': (c_i64, [c_int, c_int, c_int]),
    'mmt_text_heads_fast': (c_int, [c_int, c_int, c_int, c_int]),
    'mmt_text_heads_fwd': (c_int, [ctypes.POINTER(MmtTextHeads), c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_vp, c_vp, c_vp, ctypes.POINTER(MmtTextHeadsOpts), c_vp]),
    'mmt_text_heads_bwd': (c_int, [ctypes.POINTER(MmtTextHeads), c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(MmtTextHeadsOpts),
                                   c_vp]),
    'mmt_probe_arm': (c_int, [c_vp, c_vp, c_int]),
    'mmt_probe_count': (c_int, []),
    'mmt_probe_arm_site': (c_int, [c_int, c_vp, c_vp, c_int]),
    'mmt_probe_count_site': (c_int, [c_int]),
}

_lib = None


def lib():
  """Loads the shared library (once).  Raises if it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
          'libmmt_hip.so is missing (%s): run `python -m mmt_amd.build` (hipcc, gfx950). '
          'There is no CPU fallback for the MMT hot path.' % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
      fn.restype = res
      fn.argtypes = args
    if handle.mmt_abi_version() != 3:
      raise RuntimeError('libmmt_hip.so ABI mismatch')
    _lib = handle
  return _lib


def check(rc, what):
  if rc != 0:
    raise RuntimeError('%s failed with code %d' % (what, rc))
