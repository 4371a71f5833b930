"""The native text tower (mmt_amd.text_bert.TextBertModel) against (a) the transformers BertModel outputs and
gradients stored in tests/golden/text_bert.npz and (b) the oracle at the bert-base-cased shape the reference fine-tunes
(12 layers x 768, 12 heads x 64, 28 996 tokens; model/model.py:152-162).  bf16 MFMA operands / fp32 everything else:
hidden states atol 0.03, gradients 3 % of their norm (SURVEY 8c tolerances)."""
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


@pytest.mark.parametrize('cls_only', [False, True])
def test_text_tower_matches_transformers_golden(cls_only):
  gold, cfg, sd, ids, mask, probe = load_text_bert_fixture()
  model = _native(cfg, sd).train()  # dropout probabilities are 0 in this fixture
  model.cls_only = cls_only
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
