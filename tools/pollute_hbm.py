"""Debugging aid (GPU box): fills as much HBM as it can get with a 32-bit pattern and exits, so that a following run sees non-zero
garbage in freshly allocated device memory if the driver hands it out unscrubbed.   python tools/pollute_hbm.py 0x7fc00000"""
import torch, sys
pat = int(sys.argv[1], 0)
ts = []
try:
    for _ in range(60):
        t = torch.empty(1 << 30, dtype=torch.int32, device="cuda:0")   # 4 GiB each
        t.fill_(pat)
        ts.append(t)
except Exception as e:
    pass
torch.cuda.synchronize()
print("polluted", len(ts) * 4, "GiB with", hex(pat))
