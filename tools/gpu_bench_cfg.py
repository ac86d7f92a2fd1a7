"""Run a BASELINE.json config through the native step runner and print ONE JSON line (committed under profiles/ by
tools/run_cfg_benches.sh).   python tools/gpu_bench_cfg.py CFG [B] [STEPS] [train|eval]
CFG 2: wav2vec2-base -> bart-base ds 2;  4: hubert-large-ll60k -> mbart-large-50 ds 8;  5: SpeechMixSelf wav2vec2-large -> t5-large,
share_layer_ratio 0.5, ds 8 (LM frozen, text_input_ids [B, 33]).  Synthetic 10 s clips, 32 label tokens, bf16, Adafactor."""
import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechmix_amd.model import SpeechMixEED, SpeechMixSelf
from speechmix_amd.trainer import StepRunner

NAMES = {"2": "SpeechMixEED wav2vec2-base + bart-base, down_scale 2",
         "4": "SpeechMixEED hubert-large-ll60k + mbart-large-50, down_scale 8",
         "5": "SpeechMixSelf wav2vec2-large + t5-large, share_layer_ratio 0.5, down_scale 8 (LM frozen)"}


def build(cfg):
    with contextlib.redirect_stdout(io.StringIO()):
        if cfg == "4":
            return SpeechMixEED("hubert_large_ll60k", "facebook/mbart-large-50", down_scale=8)
        if cfg == "5":
            return SpeechMixSelf("wav2vec2_large_960", "t5-large", share_layer_ratio=0.5, down_scale=8)
        return SpeechMixEED("facebook/wav2vec2-base", "facebook/bart-base", down_scale=2)


if __name__ == "__main__":
    cfg = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    mode = sys.argv[4] if len(sys.argv) > 4 else "train"
    model = build(cfg)
    if os.environ.get("SMX_TUNE_DUMP"):
        from speechmix_amd import ops as _ops
        _ops.TUNE_LOG = []
    model.train(mode == "train")
    V = model.decoder_model.config.vocab_size
    g = torch.Generator().manual_seed(0)
    wave = (torch.randn(B, 160000, generator=g) * 0.1).clamp_(-1, 1).cuda()
    labels = torch.randint(4, V, (B, 32), generator=g).cuda()
    text = torch.randint(4, V, (B, 33), generator=g).cuda() if cfg == "5" else None
    runner = StepRunner(model, lr=1e-5, optimizer=os.environ.get("SMX_OPT", "adafactor"))
    from speechmix_amd import graphs, ops
    # set-up as in bench.py: eager steps, the capture, and in `auto` mode the timed trial of both schedules
    n_setup = graphs.WARM_STEPS + 1 + (2 * graphs.TRIAL_STEPS + 1 if graphs.MODE == "auto" else 0) if graphs.ENABLED else 3
    for i in range(n_setup):
        loss = runner.step(wave, labels, text_input_ids=text)
    torch.cuda.synchronize()
    mem = torch.cuda.max_memory_allocated() / 1e9
    t0 = time.perf_counter()
    for i in range(steps):
        loss = runner.step(wave, labels, text_input_ids=text)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # roofline object by bench.py's fixed rule (the GEMM kernel family with the largest share of the GEMM flops), from the same
    # steps repeated with HIP events around every GEMM launch (eager: events cannot be recorded inside a replayed graph)
    prof = ops.GemmProfile()
    ops.GEMM_PROFILE = prof
    for i in range(steps):
        runner.step(wave, labels, text_input_ids=text)
    torch.cuda.synchronize()
    ops.GEMM_PROFILE = None
    fams = prof.families()
    dom = max(fams, key=lambda k: fams[k]["flops"])
    fd, allg, enc = fams[dom], prof.robust(), prof.by_tag("enc_layer")
    PEAK = 2500.0
    roofline = {"kernel": dom + " (all instantiations)", "bound": "mfma", "achieved": round(fd["tflops"], 1), "peak": PEAK, "unit": "TFLOP/s",
                "frac": round(fd["tflops"] / PEAK, 4), "traffic": None, "launches_per_step": round(fd["launches"] / steps, 1),
                "avg_launch_us": round(1e3 * fd["total_ms"] / fd["launches"], 2), "flops_per_launch": round(fd["flops"] / fd["launches"]),
                "flops_share_of_gemms": round(fd["flops"] / max(allg["flops"], 1.0), 3),
                "encoder_gemms_frac": round(enc["tflops"] / PEAK, 4) if enc["launches"] else None,
                "all_gemms_frac": round(allg["tflops"] / PEAK, 4),
                "step_frac": round(allg["flops"] / steps / dt / 1e12 / PEAK, 4),
                "rule": "kernel family with the largest share of GEMM flops; launches x median per (variant, shape); step_frac: GEMM flops only"}
    if os.environ.get("SMX_TUNE_DUMP"):
        us = lambda t: f"{t*1e3:8.1f}" if t is not None else "       -"
        for rec in _ops.TUNE_LOG:
            key, t1, t8, mode, *rest = rec
            print("tune", us(t1), us(t8), *(us(r) for r in rest), "->", mode, key[:10] if isinstance(key[0], tuple) else key, file=sys.stderr)
    print(json.dumps({"config": int(cfg), "workload": NAMES[cfg], "mode": mode, "batch": B, "clip_seconds": 10.0, "steps": steps,
                      "ms_per_step": round(dt * 1e3, 2), "audio_s_per_s": round(B * 10 / dt, 1), "dtype": "bf16",
                      "optimizer": "adafactor", "params_M": round(model.store.total / 1e6, 1), "peak_mem_GB": round(mem, 1),
                      "final_loss": round(loss.item(), 4), "n_gpus": 1, "roofline": roofline,
                      "step_graphs": runner._graphs is not None, "graph_trial_fwd_bwd_ms": runner.graph_trial_ms,
                      "ffn_ld_padding": os.environ.get("SMX_PAD_FFN", "1") != "0"}), flush=True)
