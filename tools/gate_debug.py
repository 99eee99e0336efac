import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "60")), exit=True)
import torch
import bench
from vspbfr_amd import hip_ops
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(os.environ.get("B", "8"))
pipe = bench.build_pipeline(dev, 50, True, 123)
lq = hip_ops.keyed_fill([(B, 3, 512, 512)], [hip_ops.SEG_LQ], 123, 0, dist="uniform", device=dev)[0]
with torch.no_grad():
    pipe(lq)
    torch.cuda.synchronize()
    print("serial ok", flush=True)
    t0 = time.perf_counter()
    for k, o in enumerate(pipe.run_batches([(lq, i * B) for i in range(N)])):
        print("yield", k, round(time.perf_counter() - t0, 3), "tickets", pipe._gates.ticket if hasattr(pipe, "_gates") and pipe._gates else None,
              "written", pipe._gates.written if hasattr(pipe, "_gates") and pipe._gates else None, flush=True)
    print("loop enqueued", round(time.perf_counter() - t0, 3), flush=True)
    torch.cuda.synchronize()
    print("drained", round(time.perf_counter() - t0, 3), flush=True)
