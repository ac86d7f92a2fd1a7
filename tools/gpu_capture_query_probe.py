"""What another thread may do while a HIP-graph capture is open (round 5; why graphs.StepGraphs captures thread-locally and on one stream per engine).
torch.distributed's NCCL watchdog thread polls work events with hipEventQuery.  Measured on ROCm 7.0 / MI355X:
  capture mode global        -> ANY hipEventQuery from another thread fails ("operation not permitted when stream is capturing") and the capture is lost
  thread_local / relaxed     -> queries of events recorded on OTHER streams work; an event last recorded on the capturing stream still fails
  torch's stream pool        -> 32 streams, round-robin: the 33rd torch.cuda.Stream() is the first one again
"""
import torch, threading
torch.cuda.set_device(0)
x = torch.zeros(1024, device="cuda")
for mode in ("global", "thread_local", "relaxed"):
    for same_stream in (True, False):
        s = torch.cuda.Stream(); s2 = torch.cuda.Stream()
        e = torch.cuda.Event()
        with torch.cuda.stream(s if same_stream else s2):
            x.add_(1); e.record()
        torch.cuda.synchronize()
        res = {}
        def q():
            torch.cuda.set_device(0)
            try:
                res["q"] = e.query()
            except Exception as ex:
                res["q"] = "ERR " + str(ex).split("\n")[0][:80]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            g.capture_begin(capture_error_mode=mode)
            x.add_(1)
            t = threading.Thread(target=q); t.start(); t.join()
            try:
                g.capture_end(); end = "ok"
            except Exception as ex:
                end = "ERR " + str(ex).split("\n")[0][:80]
        print(f"mode {mode:12s} event recorded on the {'capturing' if same_stream else 'other'} stream: query from another thread -> {res['q']} | capture_end {end}", flush=True)
# pool wrap-around: how many Stream() objects until the handle repeats
seen = {}
for i in range(80):
    st = torch.cuda.Stream()
    if st.cuda_stream in seen:
        print("pool stream handle repeats after", i - seen[st.cuda_stream], "creations"); break
    seen[st.cuda_stream] = i
