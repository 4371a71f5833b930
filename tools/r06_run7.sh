source "$(dirname "$0")/r06_common.sh"
cd $R
S="--steps 300 --warmup 20"
ab rag_host "$S --host-inputs --ragged-inputs" "MMT_X=0"
ab rag_host_frac "$S --host-inputs --ragged-inputs" "MMT_LIVE_FRACTION=0.52"
ab rag_resident "$S --ragged-inputs" "MMT_X=0"
ab rag_resident_frac "$S --ragged-inputs" "MMT_LIVE_FRACTION=0.52"
ab host_dense "$S --host-inputs" "MMT_X=0"
python - <<'PY'
import torch, bench
from mmt_amd import synthetic
from mmt_amd.feature_store import RaggedFeatures
from mmt_amd.train_step import FlatMinibatch
bench.select_config(1)
mb, text = synthetic.make_batch(1000, bench.BATCH, synthetic.MSRVTT_MODALITIES, bench.TOKENS, max_pos=bench.MAX_POS)
rag = RaggedFeatures.from_dense(mb['features'], mb['features_t'], mb['features_ind'], mb['features_maxpool'], experts=synthetic.MSRVTT_MODALITIES, pin_memory=True)
print('host live', rag.live, 32 + sum(rag.live.values()))
m2 = {k: v for k, v in mb.items() if not k.startswith('features')}; m2['features'] = rag; m2['text'] = text.view(-1, 768)
host = FlatMinibatch(m2, 'cpu', pin_memory=True)
dev = FlatMinibatch(host, torch.device('cuda', 0))
print('dev live', dev['features'].live)
PY
