"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz from the REAL reference.

Run in the build container only (needs /root/reference):

    python -m oracle.gen_golden            # from the repo root

For every fixture it (1) builds the reference `CENet` / `BertModel` /
`MaxMarginRankingLoss` on CPU with seeded parameters and inputs
(`mmt_amd.synthetic`), (2) runs the reference forward + backward, (3) runs the
oracle restatement (`oracle.mmt_oracle`) on the same inputs and ASSERTS that it
agrees with the reference, (4) stores the reference's outputs.  Parameters and
inputs are not stored: they are regenerated from the seeds, and checksums in the
fixture detect any drift of the generator.

The text tower (HF bert-base-cased, transformers==3.1.0, requirements.txt:42;
call sites model/model.py:161-162,371-376) is a third-party dependency whose
weights are not available offline: it is replaced by a stub that returns a
seeded (B*C, 1, 768) tensor, i.e. the fixtures start at the text tower's output.
"""
import collections
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mmt_amd import synthetic  # noqa: E402
from oracle import mmt_oracle as O  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

GRAD_PROBES = [
    'vid_bert.encoder.layer.0.attention.self.query.weight',
    'vid_bert.encoder.layer.0.attention.self.value.bias',
    'vid_bert.encoder.layer.{last}.output.dense.weight',
    'vid_bert.encoder.layer.{last}.intermediate.dense.bias',
    'vid_bert.encoder.layer.{last}.output.layer_norm.weight',
    'vid_bert.embeddings.position_embeddings.weight',
    'vid_bert.embeddings.token_type_embeddings.weight',
    'vid_bert.embeddings.layer_norm.bias',
    'video_dim_reduce.{m0}.fc.weight',
    'video_dim_reduce.{m1}.fc.bias',
    'text_GU.{m0}.fc.weight',
]

FIXTURES = {
    # small everything; dh=128 like the published configs
    'tiny': dict(modalities=['ocr', 'speech', 'vggish'], batch=6, max_tokens=5,
                 vb=dict(hidden=256, layers=2, heads=2, inter=512, max_pos=32), seed=11),
    # BASELINE.json configs[0]: 2 experts, 1 BERT layer, batch 8
    'configA': dict(modalities=['s3d', 'vggish'], batch=8, max_tokens=30,
                    vb=dict(hidden=512, layers=1, heads=4, inter=3072, max_pos=32), seed=12),
    # BASELINE.json configs[1]: 7 experts x 30 tokens, d512, L4, batch 32
    'configB': dict(modalities=synthetic.MSRVTT_MODALITIES, batch=32, max_tokens=30,
                    vb=dict(hidden=512, layers=4, heads=4, inter=3072, max_pos=32), seed=13),
    # BASELINE.json configs[3] shape (long sequences): 7 experts x 100 tokens -> S = 708, max_pos 102; small batch
    'config4': dict(modalities=synthetic.MSRVTT_MODALITIES, batch=4, max_tokens=100,
                    vb=dict(hidden=512, layers=4, heads=4, inter=3072, max_pos=102), seed=14),
    # ... and the same shape at the BENCHMARK batch (32 pairs: 22 656 token rows, 177 row tiles of 128; attention scores
    # (32, 4, 708, 708)): what BASELINE.json configs[3] is timed on
    'config4b32': dict(modalities=synthetic.MSRVTT_MODALITIES, batch=32, max_tokens=100,
                       vb=dict(hidden=512, layers=4, heads=4, inter=3072, max_pos=102), seed=16),
    # BASELINE.json configs[4] encoder shape (HowTo100M-scale): d1024, 6 layers, 8 heads, I = 6144; small batch
    'config5': dict(modalities=synthetic.MSRVTT_MODALITIES, batch=4, max_tokens=10,
                    vb=dict(hidden=1024, layers=6, heads=8, inter=6144, max_pos=32), seed=15),
    # ... and the same encoder at the MSRVTT token count and a batch that fills whole GEMM tiles (16 x 218 = 3 488 token
    # rows; d1024 / 8 heads / I = 6144 / 6 layers: every GEMM shape, head count and split-K path of configs[4])
    'config5b16': dict(modalities=synthetic.MSRVTT_MODALITIES, batch=16, max_tokens=30,
                       vb=dict(hidden=1024, layers=6, heads=8, inter=6144, max_pos=32), seed=17),
}


def subsample(t, limit=4096):
  flat = t.detach().reshape(-1)
  stride = max(1, flat.numel() // limit)
  return flat[::stride][:limit].clone().numpy()


def arch_args(fx):
  vb = synthetic.vid_bert_params(dropout=0.0, **fx['vb'])
  return dict(
      l2renorm=False, keep_missing_modalities=True, test_caption_mode='indep',
      txt_inp='bertftn', txt_agg='bertftn', txt_wgh='emb', vid_wgh='none',
      vid_cont='bert', vid_inp='both', pos_enc='tint', out_tok='mxp',
      vid_bert_params=vb, txt_pro='gbn', same_dim=fx['vb']['hidden'],
      txt_bert_params={'hidden_dropout_prob': 0.0, 'attention_probs_dropout_prob': 0.0})


def build_reference_cenet(R, fx):
  class FakeTxtBert(torch.nn.Module):
    """Stands in for transformers.BertModel (model/model.py:161-162)."""

    def __init__(self):
      super().__init__()
      self.config = type('C', (), {'hidden_size': 768})()
      self.embeddings = torch.nn.Module()
      self.text = None

    @classmethod
    def from_pretrained(cls, name, **kw):
      return cls()

    def forward(self, input_ids, attention_mask=None, token_type_ids=None,
                position_ids=None, head_mask=None):
      return (self.text[:, None, :],)

  R.model.TxtBertModel = FakeTxtBert
  config = {'experts': {'modalities': fx['modalities'], 'face_dim': 512}}
  expert_dims = R.util.compute_dims(config)
  model = R.model.CENet(expert_dims=expert_dims, tokenizer=None, **arch_args(fx))
  return model, expert_dims


def run_cenet_fixture(R, name, fx):
  torch.manual_seed(0)
  model, expert_dims = build_reference_cenet(R, fx)
  shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
  sd = synthetic.make_state_dict(fx['seed'], shapes)
  model.load_state_dict(sd)
  mods = list(expert_dims.keys())
  assert mods == list(synthetic.compute_dims(fx['modalities']).keys())
  mb, text = synthetic.make_batch(fx['seed'], fx['batch'], fx['modalities'], fx['max_tokens'],
                                  max_pos=fx['vb']['max_pos'])
  loss_fn = R.loss.MaxMarginRankingLoss(margin=0.05, fix_norm=True)
  out = collections.OrderedDict()
  out['meta'] = json.dumps(dict(
      name=name, fixture=fx, modalities=mods, param_shapes={k: list(v) for k, v in shapes.items()},
      param_checksums={k: synthetic.checksum(v) for k, v in sd.items() if v.dtype.is_floating_point},
      text_checksum=synthetic.checksum(text),
      feature_checksum=synthetic.checksum(mb['features'][mods[0]])))

  def ref_forward(training, out_mode):
    model.train(training)
    m = copy.deepcopy(mb)
    t = text.clone().requires_grad_(True)
    model.txt_bert.text = t.view(-1, t.shape[-1])
    return model(m['token_ids'], m['features'], m['features_t'], m['features_ind'],
                 m['features_avgpool'], m['features_maxpool'], m['query_masks'],
                 out=out_mode, device='cpu'), t

  # --- eval mode (BatchNorm running stats) ---
  with torch.no_grad():
    ev, _ = ref_forward(False, 'conf')
    emb, _ = ref_forward(False, 'embds')
  out['eval_sims'] = ev['cross_view_conf_matrix'].numpy()
  for k in ('vid_embds', 'text_embds', 'vid_weights', 'text_weights'):
    out['eval_' + k] = emb[k].numpy()
  out['eval_t2v'] = json.dumps({k: v for k, v in R.metric.t2v_metrics(out['eval_sims'].copy()).items() if k != 'cols'})
  out['eval_v2t'] = json.dumps({k: v for k, v in R.metric.v2t_metrics(out['eval_sims'].copy()).items() if k != 'cols'})

  # --- train mode (BatchNorm batch stats, dropout p=0), forward + backward ---
  sd_before = copy.deepcopy(model.state_dict())
  model.zero_grad()
  tr, t_in = ref_forward(True, 'conf')
  sims = tr['cross_view_conf_matrix']
  loss = loss_fn(sims)
  loss.backward()
  out['train_sims'] = sims.detach().numpy()
  out['train_loss'] = np.float64(loss.item())
  out['train_text_grad'] = subsample(t_in.grad)
  grads = {n: p.grad for n, p in model.named_parameters()}
  assert grads['vid_bert.pooler.dense.weight'] is None
  last = fx['vb']['layers'] - 1
  for probe in GRAD_PROBES:
    key = probe.format(last=last, m0=mods[0], m1=mods[-1])
    g = grads[key]
    out['grad/' + key] = subsample(g)
    out['gradnorm/' + key] = np.float64(g.double().norm().item())
  bn_key = 'text_GU.%s.cg.batch_norm.running_mean' % mods[0]
  out['bn_running_mean_after'] = model.state_dict()[bn_key].numpy()
  # r04: the other two BatchNorm buffers of that text head after the train-mode forward (model/model.py:736-750)
  out['bn_running_var_after'] = model.state_dict()[bn_key.replace('running_mean', 'running_var')].numpy()
  out['bn_num_batches_tracked_after'] = np.int64(model.state_dict()[bn_key.replace('running_mean', 'num_batches_tracked')].item())

  # --- oracle vs reference (fp32 oracle; fp64 oracle bounds the fp32 noise) ---
  cfg = dict(modalities=mods, expert_dims=expert_dims, vid_bert_params=arch_args(fx)['vid_bert_params'],
             same_dim=fx['vb']['hidden'], test_caption_mode='indep')
  P = {k: v.clone() for k, v in sd_before.items()}
  with torch.no_grad():
    o_ev = O.cenet_forward(P, cfg, copy.deepcopy(mb), text, training=False)
    o_emb = O.cenet_forward(P, cfg, copy.deepcopy(mb), text, training=False, out='embds')
  err = np.abs(o_ev['cross_view_conf_matrix'].numpy() - out['eval_sims']).max()
  assert err < 2e-5, (name, 'eval sims', err)
  for k in ('vid_embds', 'text_embds', 'vid_weights', 'text_weights'):
    e = np.abs(o_emb[k].numpy() - out['eval_' + k]).max()
    assert e < 2e-5, (name, k, e)
  Pg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd_before.items()}
  tg = text.clone().requires_grad_(True)
  o_tr = O.cenet_forward(Pg, cfg, copy.deepcopy(mb), tg, training=True)
  o_loss = O.max_margin_ranking_loss(o_tr['cross_view_conf_matrix'], 0.05, True)
  o_loss.backward()
  e = np.abs(o_tr['cross_view_conf_matrix'].detach().numpy() - out['train_sims']).max()
  assert e < 2e-5, (name, 'train sims', e)
  assert abs(o_loss.item() - loss.item()) < 1e-6, (name, o_loss.item(), loss.item())
  worst = 0.0
  gmax = max(g.norm().item() for g in grads.values() if g is not None)
  for n, g in grads.items():
    if g is None:
      assert Pg[n].grad is None or Pg[n].grad.abs().max() == 0, n
      continue
    if g.norm().item() < 1e-5 * gmax and Pg[n].grad.norm().item() < 1e-5 * gmax:
      continue  # e.g. key.bias: exactly zero by softmax shift-invariance, both sides are rounding noise
    rel = (Pg[n].grad - g).norm() / g.norm().clamp_min(1e-12)
    worst = max(worst, rel.item())
    assert rel < 2e-3, (name, n, rel.item())
  rel = (tg.grad - t_in.grad).norm() / t_in.grad.norm()
  assert rel < 2e-3, (name, 'text grad', rel.item())
  assert _close_metrics(O.t2v_metrics(out['eval_sims']), json.loads(out['eval_t2v']))
  assert _close_metrics(O.v2t_metrics(out['eval_sims']), json.loads(out['eval_v2t']))
  print('%-8s oracle==reference: eval sims err %.2e, worst param-grad rel err %.2e, loss %.6f'
        % (name, err, worst, loss.item()))
  np.savez_compressed(os.path.join(GOLDEN, 'cenet_%s.npz' % name), **out)


def _close_metrics(a, b):
  return all(abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(b[k])) for k in b)


def run_bert_fixture(R):
  """Stand-alone model/bert.py BertModel (the inner drop-in boundary)."""
  import types
  vb = synthetic.vid_bert_params(hidden=256, layers=2, heads=2, inter=512, dropout=0.0)
  torch.manual_seed(0)
  model = R.bert.BertModel(types.SimpleNamespace(**vb))
  shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
  sd = synthetic.make_state_dict(21, {('vid_bert.' + k): v for k, v in shapes.items()})
  model.load_state_dict({k[len('vid_bert.'):]: v for k, v in sd.items()})
  rs = np.random.RandomState(21)
  b, s, d = 5, 37, 256
  feats = torch.from_numpy(rs.randn(b, s, d).astype(np.float32))
  mask = torch.from_numpy((rs.rand(b, s) > 0.3).astype(np.int64))
  mask[:, 0] = 1
  types_ = torch.from_numpy(rs.randint(0, 19, size=(b, s)).astype(np.int64))
  pos = torch.from_numpy(rs.randint(0, 32, size=(b, s)).astype(np.int64))
  model.eval()
  out = collections.OrderedDict()
  with torch.no_grad():
    seq, pooled = model(types_, attention_mask=mask, token_type_ids=types_, position_ids=pos, features=feats)[:2]
    seq_nopos = model(types_, attention_mask=mask, token_type_ids=types_, position_ids=None, features=feats)[0]
  P = {k: v for k, v in sd.items()}
  with torch.no_grad():
    o_seq, o_pool = O.bert_model(P, 'vid_bert.', vb, mask, types_, pos, feats, with_pooler=True)
    o_nopos = O.bert_model(P, 'vid_bert.', vb, mask, types_, None, feats)
  for a, b_ in ((o_seq, seq), (o_pool, pooled), (o_nopos, seq_nopos)):
    assert (a - b_).abs().max() < 2e-5
  out['meta'] = json.dumps(dict(vb=vb, seed=21, shape=[b, s, d],
                                param_checksums={k: synthetic.checksum(v) for k, v in sd.items()}))
  out['sequence_output'] = seq.numpy()
  out['pooled_output'] = pooled.numpy()
  out['sequence_output_nopos'] = seq_nopos.numpy()
  np.savez_compressed(os.path.join(GOLDEN, 'bert_standalone.npz'), **out)
  print('bert     oracle==reference')


TEXT_BERT = dict(vocab_size=1000, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                 max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0, layer_norm_eps=1e-12, pad_token_id=0)


def run_text_bert_fixture():
  """The text tower's oracle restatement pinned against the installed transformers BertModel (the reference pins
  transformers==3.1.0; the BERT arithmetic is the same), forward and backward, head dim 64."""
  import transformers
  cfg = transformers.BertConfig(**TEXT_BERT)
  torch.manual_seed(0)
  hf = transformers.BertModel(cfg, add_pooling_layer=True)
  shapes = {k: tuple(v.shape) for k, v in hf.state_dict().items() if v.dtype.is_floating_point}
  sd = synthetic.make_state_dict(41, {('txt_bert.' + k): v for k, v in shapes.items()})
  sd['txt_bert.embeddings.word_embeddings.weight'][0].zero_()
  hf.load_state_dict({k[len('txt_bert.'):]: v for k, v in sd.items()}, strict=False)
  b, w = 6, 30
  ids, mask = synthetic.text_token_batch(41, b, w, TEXT_BERT['vocab_size'])
  pos = torch.arange(w).unsqueeze(0).expand(b, w)
  probe = torch.from_numpy(np.random.RandomState(42).randn(b, TEXT_BERT['hidden_size']).astype(np.float32))
  hf.train()  # dropout probabilities are 0: train mode only exercises the autograd path
  seq = hf(ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids), position_ids=pos)[0]
  loss = (seq[:, 0] * probe).sum()
  loss.backward()
  grads = {('txt_bert.' + n): p.grad for n, p in hf.named_parameters() if p.grad is not None}
  vb = dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, layer_norm_eps=1e-12,
            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
  P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
  o_seq = O.text_bert_model(P, 'txt_bert.', vb, ids, mask, None, pos)
  assert (o_seq - seq).abs().max() < 2e-5, (o_seq - seq).abs().max()
  (o_seq[:, 0] * probe).sum().backward()
  for k in ('txt_bert.embeddings.word_embeddings.weight', 'txt_bert.encoder.layer.0.attention.self.query.weight',
            'txt_bert.encoder.layer.1.output.dense.weight', 'txt_bert.embeddings.LayerNorm.weight'):
    ref = grads[k]
    assert (P[k].grad - ref).norm() <= 2e-4 * ref.norm(), k
  out = collections.OrderedDict()
  out['meta'] = json.dumps(dict(cfg=TEXT_BERT, seed=41, shape=[b, w], transformers=transformers.__version__,
                                param_checksums={k: synthetic.checksum(v) for k, v in sd.items()}))
  out['sequence_output'] = seq.detach().numpy()
  out['loss'] = np.float64(loss.item())
  for k in ('embeddings.word_embeddings.weight', 'embeddings.position_embeddings.weight',
            'embeddings.LayerNorm.bias', 'encoder.layer.0.attention.self.query.weight',
            'encoder.layer.0.attention.self.value.bias', 'encoder.layer.1.intermediate.dense.weight',
            'encoder.layer.1.output.dense.weight', 'encoder.layer.1.output.LayerNorm.weight'):
    out['grad.' + k] = grads['txt_bert.' + k].numpy()
  np.savez_compressed(os.path.join(GOLDEN, 'text_bert.npz'), **out)
  print('text_bert oracle==transformers %s' % transformers.__version__)


def run_sim_loss_metric_fixtures(R):
  rs = np.random.RandomState(31)
  out = collections.OrderedDict()
  # --- similarity, multi-caption, both merge modes, with a zero-weight row (:816) ---
  b, c, m, d = 5, 3, 3, 16
  mods = ['a', 'b', 'c']
  vid = torch.from_numpy(rs.randn(b, m, d).astype(np.float32))
  txt = torch.from_numpy(rs.randn(b, m, c, d).astype(np.float32))
  vw = torch.from_numpy(rs.rand(b, m).astype(np.float32))
  tw = torch.from_numpy(rs.rand(b, c, m).astype(np.float32))
  vw[2] = 0.0  # video with every expert weight zero -> norm_weights==0 branch
  for mode in ('avg', 'indep'):
    ref = R.model.sharded_cross_view_inner_product(
        vid_embds=collections.OrderedDict((k, vid[:, i].clone()) for i, k in enumerate(mods)),
        text_embds=collections.OrderedDict((k, txt[:, i].clone()) for i, k in enumerate(mods)),
        vid_weights=vw.clone(), text_weights=tw.clone(), subspaces=mods,
        merge_caption_similiarities=mode)
    orc = O.cross_view_inner_product(vid, txt, vw, tw, mode)
    assert (ref - orc).abs().max() < 1e-5, mode
    out['sims_' + mode] = ref.numpy()
  for k, v in (('vid', vid), ('txt', txt), ('vw', vw), ('tw', tw)):
    out['sim_in_' + k] = v.numpy()
  # --- losses: known-answer vectors of SURVEY 8a + random ---
  kat = torch.tensor([[0.5, 0.6], [0.1, 0.2]])
  out['kat_x'] = kat.numpy()
  out['kat_mm_fix'] = np.float64(R.loss.MaxMarginRankingLoss(0.05, True)(kat).item())
  out['kat_mm_nofix'] = np.float64(R.loss.MaxMarginRankingLoss(0.05, False)(kat).item())
  out['kat_nce'] = np.float64(R.loss.InfoNceLoss()(kat).item())
  assert abs(out['kat_mm_fix'] - 0.15) < 1e-7 and abs(out['kat_mm_nofix'] - 0.10) < 1e-7
  assert abs(out['kat_nce'] - 1.407412) < 1e-5
  for n in (3, 17, 64):
    x = torch.from_numpy((0.3 * rs.randn(n, n)).astype(np.float32)).requires_grad_(True)
    for fix in (True, False):
      l = R.loss.MaxMarginRankingLoss(0.05, fix)(x)
      g, = torch.autograd.grad(l, x)
      xo = x.detach().clone().requires_grad_(True)
      lo = O.max_margin_ranking_loss(xo, 0.05, fix)
      go, = torch.autograd.grad(lo, xo)
      assert abs(l.item() - lo.item()) < 1e-6 and (g - go).abs().max() < 1e-6
      out['mm_%d_%d' % (n, fix)] = np.float64(l.item())
      out['mm_grad_%d_%d' % (n, fix)] = g.numpy()
    ln = R.loss.InfoNceLoss()(x)
    assert abs(ln.item() - O.info_nce_loss(x.detach()).item()) < 1e-5
    out['nce_%d' % n] = np.float64(ln.item())
    out['loss_x_%d' % n] = x.detach().numpy()
  # --- metrics incl. ties, multi-caption and query masks ---
  cases = {}
  s1 = rs.randn(40, 40).astype(np.float32)
  s2 = np.round(rs.randn(30, 30), 1).astype(np.float32)  # many ties
  s3 = rs.randn(60, 20).astype(np.float32)  # 3 captions / video
  qm = (rs.rand(20, 3) > 0.2).astype(np.float32)
  qm[:, 0] = 1
  for key, s, q in (('plain', s1, None), ('ties', s2, None), ('multi', s3, None), ('masked', s3, qm)):
    t2v = R.metric.t2v_metrics(s.copy(), None if q is None else q.copy())
    v2t = R.metric.v2t_metrics(s.copy(), None if q is None else q.copy())
    ot = O.t2v_metrics(s.copy(), q)
    ov = O.v2t_metrics(s.copy(), q)
    t2v.pop('cols'), v2t.pop('cols')
    assert _close_metrics(ot, t2v) and _close_metrics(ov, v2t), key
    out['metric_sims_' + key] = s
    if q is not None:
      out['metric_qm_' + key] = q
    cases[key] = dict(t2v={k: float(v) for k, v in t2v.items()}, v2t={k: float(v) for k, v in v2t.items()})
  out['metric_cases'] = json.dumps(cases)
  np.savez_compressed(os.path.join(GOLDEN, 'sim_loss_metric.npz'), **out)
  print('sim/loss/metric oracle==reference')


def main():
  os.makedirs(GOLDEN, exist_ok=True)
  R = load_reference()
  torch.set_num_threads(os.cpu_count())
  only = [a for a in sys.argv[1:] if not a.startswith('-')]
  if not only or 'text_bert' in only:
    run_text_bert_fixture()
  if not only:
    run_sim_loss_metric_fixtures(R)
    run_bert_fixture(R)
  for name, fx in FIXTURES.items():
    if not only or name in only:
      run_cenet_fixture(R, name, fx)


if __name__ == '__main__':
  main()
