import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vspbfr_amd import hip_ops as H
side = torch.cuda.Stream(priority=-1)
main = torch.cuda.current_stream()
sig = H.StreamSignal()
t = torch.zeros(1, device="cuda")
with torch.cuda.stream(side):
    sig.wait_geq(1)
    t.add_(1)
time.sleep(0.5)
print("before write:", side.query(), "(False = still waiting)")
sig.write(1)
torch.cuda.synchronize()
print("after write:", side.query(), float(t))
with torch.cuda.stream(side):
    sig.wait_geq(5)
    t.add_(1)
time.sleep(0.2)
print("waiting for 5:", side.query())
sig.write(3)
time.sleep(0.2)
print("after write 3:", side.query())
sig.write(7)
torch.cuda.synchronize()
print("after write 7:", side.query(), float(t))
