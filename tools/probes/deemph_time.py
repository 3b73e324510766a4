# FMDeemph<int16> alone: time per call for 1024 resident channels (usage: deemph_time.py [alpha ...]; SDRHIP_DEEMPH_SPEC=0 for the
# one-lane kernel, SDRHIP_DEEMPH_TILED=1 for the LDS-tiled one of rounds 1-2); data: a tone plus noise, or "const" rows (the
# segmented kernel's worst case: no two runs ever meet)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libsdr_amd as sa
dev = torch.device('cuda', 0)
st = torch.cuda.Stream(device=dev)
alphas = [int(v) for v in sys.argv[1:] if v.isdigit()] or [4]
const = "const" in sys.argv
with torch.cuda.stream(st):
    ctx = sa.Context(0, stream=st.cuda_stream)
    for alpha in alphas:
        for n in (524, 789, 3276, 8192):
            C = int(os.environ.get("DE_C", "1024"))
            node = sa.FMDeemphI16(ctx, alpha, channels=C, max_in=n)
            x = (3000 * torch.sin(torch.arange(n, device=dev) * 0.01)[None, :] + 200 * torch.randn((C, n), device=dev)).to(torch.int16)
            if const:
                x = x[:, :1].repeat(1, n).contiguous()
            y = torch.zeros_like(x)
            for i in range(10): node.process_dev(x.data_ptr(), n, n, y.data_ptr(), n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 200
            for i in range(K): node.process_dev(x.data_ptr(), n, n, y.data_ptr(), n)
            torch.cuda.synchronize()
            print('alpha %d%s, %d outputs per channel: %.1f us per call' % (alpha, " const" if const else "", n, (time.perf_counter() - t0) / K * 1e6))
