"""Control for graph_abort.py: a torch-only graph replayed after an eager null-stream kernel."""
import sys, torch
dev = torch.device("cuda:0")
x = torch.randn(1 << 16, device=dev, requires_grad=True)
w = torch.randn(1 << 16, device=dev, requires_grad=True)
def step():
    x.grad = None; w.grad = None
    y = (x * w).view(256, 256)
    z = torch.zeros(256, 256, device=dev)
    z += y @ y
    z.sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side): step()
for it in range(3):
    g.replay(); torch.cuda.synchronize()
print("replays OK", flush=True)
x.grad.zero_(); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize(); print("replay after eager zero OK", flush=True)
