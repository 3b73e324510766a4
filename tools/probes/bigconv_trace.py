import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import libsdr_amd as sa
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = sa.Context(0, stream=stream.cuda_stream)
    Nb, C, N = 16384, 256, 65536
    K = sa.design_fftfilt_spectrum(sa.design_fftfilt_kernel(Nb, 50e3, 150e3, 2.4e6))
    node = sa.FFTConv(ctx, sa.FFTCONV_OLA, 2 * Nb, K, channels=C, max_in=N)
    x = torch.randn((C, N, 2), dtype=torch.float32, device=dev) * 0.3
    y = torch.zeros((C, N, 2), dtype=torch.float32, device=dev)
    for _ in range(10):
        node.process_dev(x.data_ptr(), N, N, y.data_ptr(), N)
    torch.cuda.synchronize()
