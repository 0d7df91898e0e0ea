"""Diagnostic: UNet actor inference throughput (rows / s) in a few execution modes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_agent_rl_wrsn_amd import build_networks
UNet, _ = build_networks(100)
net = UNet().cuda()
x = torch.rand(512, 4, 100, 100, device="cuda")
def bench(tag, fn, n=6):
    with torch.no_grad():
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-34s %8.1f ms per 512 rows  (%.0f rows/s, %.1f TFLOP/s)" % (tag, 1e3 * dt, 512 / dt, 512 * 5.2e9 / dt / 1e12), flush=True)
bench("fp32 NCHW", lambda: net(x))
torch.backends.cudnn.benchmark = True
bench("fp32 NCHW + benchmark", lambda: net(x))
net_cl = net.to(memory_format=torch.channels_last); xcl = x.contiguous(memory_format=torch.channels_last)
bench("fp32 channels_last", lambda: net_cl(xcl))
def amp(): 
    with torch.autocast("cuda", dtype=torch.bfloat16): return net_cl(xcl)
bench("bf16 autocast channels_last", amp)
net.eval()
bench("fp32 eval-mode BN (channels_last)", lambda: net_cl(xcl))
