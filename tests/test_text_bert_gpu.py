"""The native text tower (mmt_amd.text_bert.TextBertModel) against (a) the transformers BertModel outputs and
gradients stored in tests/golden/text_bert.npz and (b) the oracle at the bert-base-cased shape the reference fine-tunes
(12 layers x 768, 12 heads x 64, 28 996 tokens; model/model.py:152-162).  bf16 MFMA operands / fp32 everything else:
hidden states atol 0.03, gradients 3 % of their norm (SURVEY 8c tolerances)."""
import types

import numpy as np
import pytest
import torch

from tests.fixtures import load_text_bert_fixture

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _native(cfg, sd):
  import types
  from mmt_amd.text_bert import TextBertModel
  model = TextBertModel(types.SimpleNamespace(hidden_act='gelu', initializer_range=0.02, **cfg))
  missing = model.load_state_dict({k[len('txt_bert.'):]: v for k, v in sd.items()}, strict=True)
  assert not missing.missing_keys and not missing.unexpected_keys
  return model.to(DEV)


def _relerr(a, b):
  return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize('cls_only,pack', [(False, False), (True, False), (True, True)])
def test_text_tower_matches_transformers_golden(cls_only, pack):
  gold, cfg, sd, ids, mask, probe = load_text_bert_fixture()
  model = _native(cfg, sd).train()  # dropout probabilities are 0 in this fixture
  model.cls_only, model.pack_tokens = cls_only, pack
  pos = torch.arange(ids.shape[1]).unsqueeze(0).expand_as(ids)
  seq = model(ids.to(DEV), attention_mask=mask.to(DEV), token_type_ids=torch.zeros_like(ids).to(DEV),
              position_ids=pos.to(DEV), head_mask=None)[0]
  ref = gold['sequence_output']
  if cls_only:
    assert tuple(seq.shape) == (ids.shape[0], 1, cfg['hidden_size'])
    assert np.abs(seq[:, 0].detach().cpu().numpy() - ref[:, 0]).max() < 0.03
  else:
    valid = mask.numpy().astype(bool)  # padded query rows are never consumed (masked as keys everywhere)
    assert np.abs(seq.detach().cpu().numpy() - ref)[valid].max() < 0.03
  loss = (seq[:, 0] * probe.to(DEV)).sum()
  assert abs(loss.item() - float(gold['loss'])) < 2e-2 * max(1.0, abs(float(gold['loss'])))
  loss.backward()
  flat = model._flat
  named = dict(model.named_parameters())
  for k in [k for k in gold.files if k.startswith('grad.')]:
    name = k[len('grad.'):].replace('.LayerNorm.', '.layer_norm.')
    got = flat.view(named[name], flat.current_grad()).detach().cpu().numpy()
    assert _relerr(got, gold[k]) < 3e-2, (k, _relerr(got, gold[k]))
  g_emb = flat.view(named['embeddings.word_embeddings.weight'], flat.current_grad())
  assert g_emb[0].abs().max().item() == 0.0  # padding_idx row


def test_text_tower_bert_base_shape_matches_oracle():
  """Full bert-base-cased architecture on a batch of 8 captions x 30 tokens against the CPU oracle (seconds)."""
  from mmt_amd import synthetic
  from mmt_amd.text_bert import _BERT_BASE_CASED
  from oracle import mmt_oracle as O
  from tests.fixtures import text_bert_shapes
  cfg = dict(_BERT_BASE_CASED, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
  cfg.pop('hidden_act'), cfg.pop('initializer_range')
  sd = synthetic.make_state_dict(43, {('txt_bert.' + k): v for k, v in text_bert_shapes(cfg).items()})
  sd['txt_bert.embeddings.word_embeddings.weight'][0].zero_()
  b, w = 8, 30
  ids, mask = synthetic.text_token_batch(43, b, w, cfg['vocab_size'])
  probe = torch.from_numpy(np.random.RandomState(44).randn(b, cfg['hidden_size']).astype(np.float32))
  pos = torch.arange(w).unsqueeze(0).expand(b, w)
  probes = ['txt_bert.embeddings.word_embeddings.weight', 'txt_bert.encoder.layer.0.attention.self.query.weight',
            'txt_bert.encoder.layer.11.output.dense.weight', 'txt_bert.encoder.layer.5.intermediate.dense.bias']
  P = {k: (v.clone().requires_grad_(True) if k in probes else v) for k, v in sd.items()}
  torch.set_num_threads(min(32, torch.get_num_threads()))
  o_seq = O.text_bert_model(P, 'txt_bert.', cfg, ids, mask, None, pos)
  (o_seq[:, 0] * probe).sum().backward()
  model = _native(cfg, sd).train()
  model.cls_only = True
  seq = model(ids.to(DEV), attention_mask=mask.to(DEV), token_type_ids=None, position_ids=pos.to(DEV))[0]
  err = (seq[:, 0].detach().cpu() - o_seq[:, 0].detach()).abs().max().item()
  assert err < 0.05, err  # 12 layers of bf16 operand rounding on unit-scale hidden states
  (seq[:, 0] * probe.to(DEV)).sum().backward()
  flat, named = model._flat, dict(model.named_parameters())
  for k in probes:
    name = k[len('txt_bert.'):].replace('.LayerNorm.', '.layer_norm.')
    got = flat.view(named[name], flat.current_grad()).detach().cpu().numpy()
    assert _relerr(got, P[k].grad.numpy()) < 5e-2, (k, _relerr(got, P[k].grad.numpy()))


def test_cenet_with_native_text_tower_matches_oracle():
  """End to end with the text tower inside the model (txt_inp/txt_agg = 'bertftn': fine-tuned, model/model.py:349-379):
  token ids -> native text tower -> text heads -> similarity -> loss, gradients into the text tower and the video side,
  against the oracle (text_bert_model feeding cenet_forward) on the 'tiny' fixture's video inputs."""
  import copy
  import json
  import types
  from mmt_amd import synthetic
  from mmt_amd.loss import MaxMarginRankingLoss
  from mmt_amd.model import CENet
  from mmt_amd.text_bert import TextBertModel, bert_base_cased_config
  from oracle import mmt_oracle as O
  from tests.fixtures import load_cenet_fixture, text_bert_shapes
  fx = load_cenet_fixture('tiny')
  f = fx.meta['fixture']
  tcfg = dict(vocab_size=28996, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1024,
              max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
              layer_norm_eps=1e-12, pad_token_id=0)
  txt = TextBertModel(bert_base_cased_config(**tcfg))
  vb = synthetic.vid_bert_params(dropout=0.0, **f['vb'])
  model = CENet(l2renorm=False, expert_dims=synthetic.compute_dims(f['modalities']), tokenizer=None,
                keep_missing_modalities=True, test_caption_mode='indep', txt_inp='bertftn', txt_agg='bertftn',
                txt_wgh='emb', vid_wgh='none', vid_cont='bert', vid_inp='both', pos_enc='tint', out_tok='mxp',
                vid_bert_params=vb, txt_pro='gbn', same_dim=f['vb']['hidden'],
                txt_bert_params={'hidden_dropout_prob': 0.0, 'attention_probs_dropout_prob': 0.0}, txt_bert=txt)
  assert model._native_text_tower and model.txt_bert.cls_only
  tsd = synthetic.make_state_dict(45, {('txt_bert.' + k): v for k, v in text_bert_shapes(tcfg).items()})
  tsd['txt_bert.embeddings.word_embeddings.weight'][0].zero_()
  sd = dict(fx.state_dict)
  sd.update(tsd)
  model.load_state_dict(sd)
  model.to(DEV).train()
  mb = fx.batch
  dev_mb = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV)) for k, v in mb.items()}
  out = model(dev_mb['token_ids'], dev_mb['features'], dev_mb['features_t'], dev_mb['features_ind'],
              dev_mb['features_avgpool'], dev_mb['features_maxpool'], dev_mb['query_masks'], out='conf', device=DEV)
  sims = out['cross_view_conf_matrix']
  loss = MaxMarginRankingLoss(0.05, True)(sims)
  loss.backward()
  # ---- oracle ----
  probes = ['txt_bert.embeddings.word_embeddings.weight', 'txt_bert.encoder.layer.0.attention.self.value.weight',
            'txt_bert.encoder.layer.1.output.dense.weight', 'vid_bert.encoder.layer.0.attention.self.query.weight',
            'text_GU.%s.fc.weight' % fx.cfg['modalities'][0]]
  P = {k: (v.clone().requires_grad_(True) if k in probes else v.clone()) for k, v in sd.items()}
  tok = mb['token_ids']
  b, c, w, _ = tok.shape
  ids, mask = tok.view(b * c, w, 2)[:, :, 0].long(), tok.view(b * c, w, 2)[:, :, 1].long()
  pos = torch.arange(w).unsqueeze(0).expand(b * c, w)
  text = O.text_bert_model(P, 'txt_bert.', tcfg, ids, mask, None, pos)[:, 0].view(b, c, -1)
  o = O.cenet_forward(P, fx.cfg, copy.deepcopy(mb), text, training=True)
  o_sims = o['cross_view_conf_matrix']
  o_loss = O.max_margin_ranking_loss(o_sims, 0.05, True)
  o_loss.backward()
  assert (sims.detach().cpu() - o_sims.detach()).abs().max().item() < 3e-3
  assert abs(loss.item() - o_loss.item()) < 2e-2 * max(abs(o_loss.item()), 1e-3)
  named = dict(model.named_parameters())
  for k in probes:
    flat = model.txt_bert._flat if k.startswith('txt_bert.') else model._flat
    key = named[k.replace('.LayerNorm.', '.layer_norm.')]
    got = flat.view(key, flat.current_grad()).detach().cpu().numpy()
    ref = P[k].grad.numpy()
    cos = float((got.ravel().astype(np.float64) @ ref.ravel().astype(np.float64)) / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
    assert cos > 0.995 and abs(np.linalg.norm(got) / np.linalg.norm(ref) - 1.0) < 0.05, (k, cos)  # same bar as the video side


def test_text_token_packing_is_exact_including_dropout():
  """Dropping the padded tokens (mmt_text_plan) changes nothing: same [CLS] outputs and gradients as the padded run, with
  dropout ON (the RNG is indexed by the dense coordinate b*W + t)."""
  gold, cfg, sd, ids, mask, probe = load_text_bert_fixture()
  cfg = dict(cfg, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
  outs = []
  for pack in (False, True):
    model = _native(cfg, sd).train()
    model.cls_only, model.pack_tokens = True, pack
    model._ensure_ready(DEV)
    model._seed_dev.fill_(1234)
    seq = model(ids.to(DEV), attention_mask=mask.to(DEV), token_type_ids=None, position_ids=None)[0]
    (seq[:, 0] * probe.to(DEV)).sum().backward()
    flat = model._flat
    outs.append((seq.detach().cpu(), flat.current_grad().detach().cpu().clone()))
  assert (outs[0][0] - outs[1][0]).abs().max().item() < 2e-3
  g0, g1 = outs[0][1], outs[1][1]
  assert (g0 - g1).norm().item() < 2e-3 * g0.norm().item()


def test_text_plan_keeps_the_cls_token_of_a_fully_masked_caption():
  """A caption whose attention mask is all zero still owns its [CLS] row (cls_rows must never point into the next sample
  or past the live rows); other captions are unaffected."""
  gold, cfg, sd, ids, mask, probe = load_text_bert_fixture()
  model = _native(cfg, sd).eval()
  model.cls_only, model.pack_tokens = True, True
  with torch.no_grad():
    ref = model(ids.to(DEV), attention_mask=mask.to(DEV))[0][:, 0].clone()
    m2 = mask.clone()
    m2[1] = 0
    out = model(ids.to(DEV), attention_mask=m2.to(DEV))[0][:, 0]
  keep = [i for i in range(ids.shape[0]) if i != 1]
  assert torch.isfinite(out).all()
  assert (out[keep] - ref[keep]).abs().max().item() < 1e-5
  assert (out[1] - ref[1]).abs().max().item() > 1e-3      # its own row: [CLS] attending to itself only


def test_standalone_bert_rejects_ids_outside_the_tables():
  from mmt_amd import synthetic
  from mmt_amd.bert import BertModel
  cfg = types.SimpleNamespace(**synthetic.vid_bert_params(hidden=256, layers=1, heads=2, inter=512, max_pos=16))
  model = BertModel(cfg).to(DEV).eval()
  feats = torch.randn(2, 5, 256, device=DEV)
  pos = torch.full((2, 5), 16, device=DEV, dtype=torch.long)   # table has rows 0..15
  with pytest.raises(IndexError):
    model(None, features=feats, position_ids=pos)
  typ = torch.full((2, 5), cfg.type_vocab_size, device=DEV, dtype=torch.long)
  with pytest.raises(IndexError):
    model(None, features=feats, token_type_ids=typ)
