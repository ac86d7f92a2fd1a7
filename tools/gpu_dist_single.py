"""Exercise the RCCL gradient path on ONE GPU: a 1-rank NCCL group with the collectives forced on (side stream, stage
events, bucket ranges), compared with the collective-free step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
from speechmix_amd.model import SpeechMixEED
from speechmix_amd.trainer import StepRunner
from tests.golden_util import load_case

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
sd, inp, gold, m = load_case("eed_w2v2_bart")
res = []
for force in (False, True):
    model = SpeechMixEED(m["enc_cfg"], m["lm_cfg"], down_scale=2, compute_dtype="fp32").eval()
    model.load_state_dict(sd, strict=False)
    r = StepRunner(model, lr=1e-3, optimizer="adamw", force_comm=force)
    losses = [r.step(inp["input_values"], inp["labels"]).item() for _ in range(4)]
    res.append((losses, model.store.grad.clone(), model.store.master.clone()))
    print("force_comm", force, "active", r.reducer.active, "losses", [round(x, 5) for x in losses])
dg = (res[0][1] - res[1][1]).abs().max().item()
dp = (res[0][2] - res[1][2]).abs().max().item()
print(f"max |dgrad| {dg:.3e} (grad max {res[0][1].abs().max().item():.3e})  max |dparam| {dp:.3e}")
assert max(abs(a - b) for a, b in zip(res[0][0], res[1][0])) < 1e-5, "losses differ"
assert dg < 1e-5 * max(1.0, res[0][1].abs().max().item()) and dp < 1e-5      # run-to-run noise of atomic reductions only
dist.destroy_process_group()
print("DIST SINGLE OK")
