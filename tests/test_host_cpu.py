"""CPU tests of the host logic: ABI coverage, reference-compatible state dict, flat parameter storage."""
import ctypes
import json
import os
import re

import pytest
import torch

from tests.fixtures import load_npz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
  src = open(os.path.join(ROOT, 'include', 'mmt_hip.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  out = {}
  for m in re.finditer(r'\b(?:int|int64_t|const char\*)\s+(mmt_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
    args = m.group(2).strip()
    out[m.group(1)] = 0 if args == 'void' else len([a for a in args.split(',') if a.strip()])
  return out


def _header_define(name):
  src = open(os.path.join(ROOT, 'include', 'mmt_hip.h')).read()
  return int(re.search(r'#define\s+%s\s+(\d+)' % name, src).group(1))


def test_library_exports_every_declared_symbol_and_bindings_match():
  from mmt_amd import _lib
  decl = _header_functions()
  assert len(decl) >= 30
  handle = ctypes.CDLL(_lib.LIB_PATH)  # loading needs no GPU
  for name, nargs in decl.items():
    assert hasattr(handle, name), 'declared in mmt_hip.h but not exported: ' + name
    assert name in _lib.SIGNATURES, 'no ctypes signature for ' + name
    assert len(_lib.SIGNATURES[name][1]) == nargs, 'arity mismatch for ' + name
  assert set(_lib.SIGNATURES) == set(decl)
  assert _lib.lib().mmt_abi_version() == 3
  # struct layouts agree with the C side (sizes are what the kernels are compiled against)
  assert ctypes.sizeof(_lib.MmtEpilogue) == 144  # + dot_src / lddot / dot_out (r04), rider / rider_limit / rider_slot / live_rows_hint (r06)
  assert ctypes.sizeof(_lib.MmtPackItem) == 48
  assert ctypes.sizeof(_lib.MmtExpertIO) == 96
  assert ctypes.sizeof(_lib.MmtBertLayer) == 28 * 8
  assert ctypes.sizeof(_lib.MmtBertBatch) == 128  # + side_stream (r03), rider / rider_limits / rider_slot0 / live_rows_hint (r06)
  assert ctypes.sizeof(_lib.MmtAdamQueue) == 416 and _header_define('MMT_RIDER_SLOTS') == _lib.RIDER_SLOTS
  assert _header_define('MMT_RIDER_STAGES') == _lib.RIDER_STAGES and _lib.RIDER_STATE_WORDS == 72 + 1 + 1024 + 64 + 4096
  assert ctypes.sizeof(_lib.MmtVideoFront) == 8 + 6 * 4 + 11 * 8  # experts | M B T pack max_pos do_cast | 11 pointers
  assert ctypes.sizeof(_lib.MmtTextHeadsOpts) == 16 + 4 * 8          # + video_front (r03)


def test_gemm_tile_policy_is_a_function_of_shape_and_live_rows():
  """mmt_gemm_select_tile (gemm.hip: select_tile) for the five shipped policies -- 14: 128x128 two blocks per CU, 18: phased
  128x64 while ONE round covers the live tiles, 13: 8-wave 128x64 at two blocks per CU, 24: persistent wave-specialised kernel
  from >= 4 tiles per CU (wide) / >= 200 tiles of 128x128 (long K, narrow), 21: 256x256 from a chip's worth of such tiles, 2:
  the 4-wave dense N = 512 kernel -- as a function of (epilogue, M, N, K, packed, live rows): the live count the HOST passes
  (MmtBertBatch.live_rows_hint) decides, no constant tied to the benchmark generator's fill;
  without a hint a packed batch is priced at its allocated rows."""
  from mmt_amd import _lib
  if any(k.startswith('MMT_TILE_') or k == 'MMT_LIVE_FRACTION' for k in os.environ):
    pytest.skip('tile policy switches set in the environment')
  f = ctypes.CDLL(_lib.LIB_PATH).mmt_gemm_select_tile
  E = _lib.EPI
  shapes = [('BIAS_BF16', 1536, 512, 0), ('BIAS_DROP_RES', 512, 512, 0), ('BIAS_GELU', 3072, 512, 0), ('BIAS_DROP_RES', 512, 3072, 0),
            ('DGELU', 3072, 512, 0), ('ADD_F32', 512, 3072, 0), ('BF16', 512, 512, 1), ('ADD_F32', 512, 1536, 0)]
  #            QKV  attn-out FFN-up FFN-down dGELU dFFN-up dO  dQKV     (config B: 32 x 218 = 6976 token rows, d 512, I 3072)
  want = {(6976, 1, 3639): [14, 18, 14, 18, 14, 18, 18, 18],   # the benchmark's fill: 29 live row tiles, 232 narrow tiles = one round
          (6976, 1, 1900): [14, 18, 14, 18, 14, 18, 18, 18],   # fill 0.25
          (6976, 1, 5300): [14, 13, 14, 13, 14, 13, 13, 13],   # fill 0.75: 336 narrow tiles > one round -> two blocks per CU
          (6976, 1, 6976): [14, 13, 24, 24, 24, 24, 13, 24],   # fill 1.0 = what a packed batch without a hint is priced at
          (6976, 1, 0):    [14, 13, 24, 24, 24, 24, 13, 24],
          (6976, 0, 0):    [14, 2, 24, 24, 24, 24, 13, 24],    # dense rows
          (22656, 1, 11600): [24, 13, 24, 24, 24, 24, 13, 24]}  # configs[3], S = 708
  for (M, packed, live), tiles in want.items():
    got = [f(E[e], M, n, k, packed, live, 0, dot, 0) for e, n, k, dot in shapes]
    assert got == tiles, ((M, packed, live), got)
  # configs[4] (d 1024, I 6144, 14.5 k live of 27.9 k rows): the 256x256 kernel; short batches and the compact last layer: tile 13
  assert [f(E['BIAS_GELU'], 27904, 6144, 1024, 1, 14464, 0, 0, 0), f(E['BIAS_DROP_RES'], 27904, 1024, 6144, 1, 14464, 0, 0, 0)] == [21, 21]
  assert [f(E['BIAS_GELU'], 960, 3072, 768, 1, 560, 0, 0, 0), f(E['BIAS_GELU'], 224, 3072, 512, 0, 0, 0, 0, 0)] == [13, 13]
  # a hint larger than the allocation or <= 0 is "unknown"; forced tiles (MmtEpilogue.reserved) win
  assert f(E['BIAS_DROP_RES'], 6976, 512, 3072, 1, 10 ** 6, 0, 0, 0) == f(E['BIAS_DROP_RES'], 6976, 512, 3072, 1, -5, 0, 0, 0) == 24
  assert f(E['BIAS_GELU'], 6976, 3072, 512, 1, 3639, 0, 0, 13) == 13 and f(E['BIAS_GELU'], 6976, 3072, 512, 0, 0, 0, 0, 1) == 1
  assert f(99, 1, 1, 1, 0, 0, 0, 0, 0) < 0


def test_product_path_refuses_cpu_tensors():
  from mmt_amd import ops
  from mmt_amd.loss import MaxMarginRankingLoss
  with pytest.raises(RuntimeError):
    ops.gemm_nt(torch.zeros(128, 64), torch.zeros(64, 64), torch.zeros(128, 64))
  with pytest.raises(RuntimeError):
    MaxMarginRankingLoss(0.05)(torch.zeros(4, 4))


def _fake_txt_bert():
  class Fake(torch.nn.Module):
    def __init__(self):
      super().__init__()
      self.config = type('C', (), {'hidden_size': 768})()
      self.embeddings = torch.nn.Module()
      self.text = None

    def forward(self, input_ids, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None):
      return (self.text[:, None, :],)
  return Fake()


def build_native_cenet(meta, pack_tokens=True, dropout=0.0):
  from mmt_amd import synthetic
  from mmt_amd.model import CENet
  fx = meta['fixture']
  vb = synthetic.vid_bert_params(dropout=dropout, **fx['vb'])
  return CENet(l2renorm=False, expert_dims=synthetic.compute_dims(fx['modalities']), tokenizer=None,
               keep_missing_modalities=True, test_caption_mode='indep', txt_inp='bertftn', txt_agg='bertftn',
               txt_wgh='emb', vid_wgh='none', vid_cont='bert', vid_inp='both', pos_enc='tint', out_tok='mxp',
               vid_bert_params=vb, txt_pro='gbn', same_dim=fx['vb']['hidden'],
               txt_bert_params={'hidden_dropout_prob': dropout, 'attention_probs_dropout_prob': dropout},
               txt_bert=_fake_txt_bert(), pack_tokens=pack_tokens)


@pytest.mark.parametrize('name', ['tiny', 'configB'])
def test_state_dict_is_reference_compatible(name):
  """Parameter names and shapes equal the REAL reference CENet's (recorded by gen_golden.py)."""
  meta = json.loads(str(load_npz('cenet_' + name)['meta']))
  model = build_native_cenet(meta)
  ours = {k: list(v.shape) for k, v in model.state_dict().items()}
  assert ours == meta['param_shapes']


def test_unsupported_modes_fail_loudly():
  meta = json.loads(str(load_npz('cenet_tiny')['meta']))
  from mmt_amd import synthetic
  from mmt_amd.model import CENet
  fx = meta['fixture']
  with pytest.raises(NotImplementedError):
    CENet(l2renorm=False, expert_dims=synthetic.compute_dims(fx['modalities']), tokenizer=None,
          keep_missing_modalities=True, test_caption_mode='indep', txt_inp='bertftn', txt_agg='bertftn',
          txt_wgh='emb', vid_wgh='none', vid_cont='coll', vid_inp='both', pos_enc='tint', out_tok='mxp',
          vid_bert_params=synthetic.vid_bert_params(**fx['vb']), txt_pro='gbn', same_dim=256,
          txt_bert=_fake_txt_bert())


def test_expert_type_ids_must_fit_the_type_table():
  """The reference's nn.Embedding raises IndexError for a type id outside the table (model/bert.py:96-99); the fused
  embedding kernel would read out of bounds, so the constructor checks."""
  meta = json.loads(str(load_npz('cenet_tiny')['meta']))
  from mmt_amd import synthetic
  from mmt_amd.model import CENet
  fx = meta['fixture']
  dims = synthetic.compute_dims(fx['modalities'])
  vb = dict(synthetic.vid_bert_params(**fx['vb']), type_vocab_size=max(e['idx'] for e in dims.values()))
  with pytest.raises(IndexError):
    CENet(l2renorm=False, expert_dims=dims, tokenizer=None, keep_missing_modalities=True, test_caption_mode='indep',
          txt_inp='bertftn', txt_agg='bertftn', txt_wgh='emb', vid_wgh='none', vid_cont='bert', vid_inp='both',
          pos_enc='tint', out_tok='mxp', vid_bert_params=vb, txt_pro='gbn', same_dim=fx['vb']['hidden'],
          txt_bert=_fake_txt_bert())


def test_flat_params_survive_load_and_move():
  from mmt_amd import synthetic
  meta = json.loads(str(load_npz('cenet_tiny')['meta']))
  model = build_native_cenet(meta)
  flat = model._flat
  before = {k: v.clone() for k, v in model.state_dict().items()}
  assert flat.ensure('cpu') is True and flat.is_flat()
  for k, v in model.state_dict().items():
    assert torch.equal(v, before[k]), k
  # q, k, v weights of a layer are one contiguous [3d, d] block (fused QKV GEMM)
  att = model.vid_bert.encoder.layer[0].attention.self
  d = att.query.weight.shape[0]
  o = flat.offset(att.query.weight)
  assert flat.offset(att.key.weight) == o + d * d and flat.offset(att.value.weight) == o + 2 * d * d
  fused = flat.master[o:o + 3 * d * d].view(3 * d, d)
  assert torch.equal(fused[d:2 * d], att.key.weight.data)
  # load_state_dict copies INTO the views: still flat, values updated
  sd = synthetic.make_state_dict(5, {k: tuple(v.shape) for k, v in model.state_dict().items()})
  model.load_state_dict(sd)
  assert flat.is_flat() and flat.ensure('cpu') is False
  assert torch.equal(fused[:d], sd['vid_bert.encoder.layer.0.attention.self.query.weight'])
  # module-level moves re-allocate .data: detected and re-flattened with values preserved
  model.double().float()
  assert not flat.is_flat()
  assert flat.ensure('cpu') is True and flat.is_flat()
  assert torch.equal(att.key.weight.data, sd['vid_bert.encoder.layer.0.attention.self.key.weight'])
  # the pooler is not an engine parameter (never receives gradients)
  assert all(not n.startswith('vid_bert.pooler') for n in flat.names)


def test_grad_regions_tile_the_flat_layout_in_backward_order():
  """The staged data-parallel backward reduces one contiguous span of the flat gradient buffer per stage: the spans
  must tile the whole layout, back to front (text heads + top layer first, expert projections + embeddings + layer 0
  last), with every parameter in exactly one of them."""
  meta = json.loads(str(load_npz('cenet_configB')['meta']))
  model = build_native_cenet(meta)
  f = model._flat
  regions = model.grad_regions()
  names = [n for n, _ in regions]
  n_layers = model.vid_bert.config.num_hidden_layers
  assert names == ['top'] + ['layer%d' % l for l in range(n_layers - 2, 0, -1)] + ['bottom']
  spans = sorted(s for _, s in regions)
  assert spans[0][0] == 0
  for (o0, c0), (o1, _) in zip(spans, spans[1:]):
    assert o0 + c0 == o1
  assert spans[-1][0] + spans[-1][1] == f.count
  offs = [off for _, (off, _) in regions]
  assert offs == sorted(offs, reverse=True)  # backward order = back to front
  top = dict(regions)['top']
  w = model.text_GU[model.modalities[0]].fc.weight
  assert top[0] <= f.offset(w) < top[0] + top[1]
  bottom = dict(regions)['bottom']
  for p in (model.video_dim_reduce[model.modalities[0]].fc.weight, model.vid_bert.embeddings.position_embeddings.weight,
            model.vid_bert.encoder.layer[0].output.dense.weight):
    assert bottom[0] <= f.offset(p) < bottom[0] + bottom[1]


def test_split_bottom_regions_tile_the_layout_too():
  """grad_regions(split_bottom=True): layer 0 is its own span, reduced before the embedding / token stage runs; the spans
  still tile the layout back to front."""
  meta = json.loads(str(load_npz('cenet_configB')['meta']))
  model = build_native_cenet(meta)
  f = model._flat
  regions = model.grad_regions(split_bottom=True)
  names = [n for n, _ in regions]
  n_layers = model.vid_bert.config.num_hidden_layers
  assert names == ['top'] + ['layer%d' % l for l in range(n_layers - 2, -1, -1)] + ['bottom']
  spans = sorted(s for _, s in regions)
  assert spans[0][0] == 0 and spans[-1][0] + spans[-1][1] == f.count
  for (o0, c0), (o1, _) in zip(spans, spans[1:]):
    assert o0 + c0 == o1
  offs = [off for _, (off, _) in regions]
  assert offs == sorted(offs, reverse=True)
  layer0, bottom = dict(regions)['layer0'], dict(regions)['bottom']
  p0 = model.vid_bert.encoder.layer[0].output.dense.weight
  assert layer0[0] <= f.offset(p0) < layer0[0] + layer0[1]
  for p in (model.video_dim_reduce[model.modalities[0]].fc.weight, model.vid_bert.embeddings.position_embeddings.weight):
    assert bottom[0] <= f.offset(p) < bottom[0] + bottom[1]
  assert bottom[1] + layer0[1] == dict(model.grad_regions())['bottom'][1]


def test_text_tower_state_dict_uses_huggingface_names():
  """bert-base-cased checkpoints (and trained MMT checkpoints, whose text tower is a transformers BertModel) must load:
  same keys and shapes as transformers' BertModel, `LayerNorm` spelling included."""
  from mmt_amd.text_bert import TextBertModel, bert_base_cased_config
  from tests.fixtures import text_bert_shapes
  cfg = dict(vocab_size=120, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
             max_position_embeddings=40, type_vocab_size=2)
  model = TextBertModel(bert_base_cased_config(**cfg))
  ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
  assert ours == text_bert_shapes(cfg)
  sd = {k: torch.randn(*s) for k, s in ours.items()}
  model.load_state_dict(sd)
  assert torch.equal(model.embeddings.layer_norm.weight, sd['embeddings.LayerNorm.weight'])
  assert [n for n, _ in model.flat_named_params()][0] == 'embeddings.word_embeddings.weight'
  with pytest.raises(RuntimeError):
    model(torch.zeros(2, 5, dtype=torch.long))  # CPU tensors: no fallback


def test_flat_minibatch_layout_is_device_independent():
  """A pinned host FlatMinibatch and a device one built from the same dict share one layout (upload = one copy)."""
  from mmt_amd import synthetic
  from mmt_amd.train_step import FlatMinibatch
  mb, text = synthetic.make_batch(3, 4, ['ocr', 'speech'], 5)
  mb['text'] = text.view(-1, 768)
  a = FlatMinibatch(mb, 'cpu')
  b = FlatMinibatch(a, 'cpu')
  assert a.flat.numel() == b.flat.numel() and torch.equal(a.flat, b.flat)
  assert torch.equal(b['features']['ocr'], mb['features']['ocr'])
  assert b['features']['ocr'].data_ptr() >= b.flat.data_ptr()


def test_bench_refuses_captured_collectives_on_several_gpus():
  """bench.py --capture-collectives was only ever exercised on a 1-rank RCCL group: with --gpus > 1 it must refuse (and say how
  to override) BEFORE any rank is spawned or any GPU is touched, so that a multi-GPU driver run cannot pick the unvalidated path."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ)
  env.pop('MMT_ALLOW_CAPTURED_COLLECTIVES', None)
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--capture-collectives'], env=env,
                     capture_output=True, text=True, timeout=300)
  out = r.stdout + r.stderr
  assert 'refusing --gpus 2' in out and 'MMT_ALLOW_CAPTURED_COLLECTIVES' in out
  assert '"metric"' not in out  # no result line


def test_persistent_gemm_tile_ownership_is_an_exact_cover(tmp_path):
  """mmt_amd/csrc/g5_own.h (tile order + which block owns which tile of gemm5.hip's persistent launches), compiled for the
  host with hipcc and swept over tile-grid shapes and launch sizes: every live tile is owned exactly once, maps to a distinct
  in-range (row, column), and the last partial round never gives an XCD more than an eighth (rounded up) of what is left --
  the r05 mapping left half the XCDs idle on the packed long-K GEMMs."""
  import shutil
  import subprocess
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  if not os.path.exists(hipcc):
    pytest.skip('hipcc not found')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  exe = str(tmp_path / 'g5_own_check')
  r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', os.path.join(root, 'tests', 'helpers', 'g5_own_check.hip'),
                      '-o', exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=str(tmp_path))
  assert r.returncode == 0, r.stdout[-2000:]
  r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
  assert r.returncode == 0 and r.stdout.startswith('ok '), r.stdout[-2000:]
