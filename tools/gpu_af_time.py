"""Event timing of the fused Adafactor step alone on config 2's parameter layout (235 M parameters, 458 tensors), global-norm
clipping on - for A/B switches read at load time (SMX_AF_GRID)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
with contextlib.redirect_stdout(io.StringIO()):
    model = SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2, compute_dtype="bf16", init_seed=0)
r = StepRunner(model, lr=5e-4, optimizer="adafactor", max_grad_norm=1.0)
st = model.store
st.grad.normal_(0, 0.01)
sh = None if st.shadow is st.master else st.shadow
def step():
    r.af.step(st.master, st.grad, sh, None, 5e-4, grad_scale=1.0, max_grad_norm=1.0)
for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record(); torch.cuda.synchronize()
print(f"SMX_AF_GRID={os.environ.get('SMX_AF_GRID', 'default')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per step ({r.af.ntiles} tiles)")
