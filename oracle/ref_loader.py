"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference (gabeur/mmt) in this container.

`/root/reference` exists only in the build container, never on the GPU box, so
this module is used exclusively by `oracle/gen_golden.py` (fixture generation)
and by CPU tests that are skipped when the reference tree is absent.  Nothing
under `mmt_amd/` may import it.

The reference's import chain pulls a few packages that are not installed here
(SURVEY.md section 8c): they are replaced by inert stubs.  `transformers` must be
imported BEFORE the stubs are installed (accelerate probes tensorboardX).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MMT_REFERENCE_ROOT", "/root/reference")


def reference_available():
  return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "bert.py"))


def _stub(name, **attrs):
  mod = types.ModuleType(name)
  mod.__dict__.update(attrs)
  mod.__path__ = []  # behaves as a package for `import a.b`
  sys.modules[name] = mod
  return mod


_loaded = {}


def load_reference():
  """Returns a namespace with the reference's model/bert, model/model, model/loss, model/metric."""
  if _loaded:
    return types.SimpleNamespace(**_loaded)
  if not reference_available():
    raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
  sys.dont_write_bytecode = True  # the reference dir is read-only
  import numpy as np
  if not hasattr(np, "bool"):
    np.bool = bool  # model/metric.py:138 uses the removed alias
  import transformers  # noqa: F401  (before stubs)
  import transformers.models.bert.modeling_bert as hf_bert
  sys.modules.setdefault("transformers.modeling_bert", hf_bert)

  class _Dummy:  # SummaryWriter / LinearWarmup stand-in
    def __init__(self, *a, **k):
      pass

    def __getattr__(self, name):
      return lambda *a, **k: None

  for name, attrs in [
      ("tensorboardX", dict(SummaryWriter=_Dummy)),
      ("typeguard", dict(typechecked=lambda f=None, **k: f if f else (lambda g: g))),
      ("ipdb", dict(set_trace=lambda *a, **k: None)),
      ("h5py", {}),
      ("pytorch_warmup", dict(LinearWarmup=_Dummy)),
      ("dominate", {}),
      ("dominate.tags", {}),
      ("gensim", {}),
      ("gensim.models", {}),
      ("gensim.models.keyedvectors", dict(KeyedVectors=_Dummy)),
      ("gensim.scripts", {}),
      ("gensim.scripts.glove2word2vec", dict(glove2word2vec=lambda *a, **k: None)),
  ]:
    if name not in sys.modules:
      _stub(name, **attrs)
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)
  _loaded["bert"] = importlib.import_module("model.bert")
  _loaded["loss"] = importlib.import_module("model.loss")
  _loaded["metric"] = importlib.import_module("model.metric")
  _loaded["model"] = importlib.import_module("model.model")
  _loaded["util"] = importlib.import_module("utils.util")
  return types.SimpleNamespace(**_loaded)
