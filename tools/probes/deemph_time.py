import sys; sys.path.insert(0,'/root/repo')
import torch, time
import libsdr_amd as sa
dev=torch.device('cuda',0)
st=torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    ctx=sa.Context(0, stream=st.cuda_stream)
    for n in (524, 789, 3276, 8192):
        C=1024
        node=sa.FMDeemphI16(ctx, 4, channels=C, max_in=n)
        x=torch.randint(-8000,8000,(C,n),dtype=torch.int16,device=dev); y=torch.zeros_like(x)
        for i in range(10): node.process_dev(x.data_ptr(), n, n, y.data_ptr(), n)
        torch.cuda.synchronize()
        t0=time.perf_counter()
        K=200
        for i in range(K): node.process_dev(x.data_ptr(), n, n, y.data_ptr(), n)
        torch.cuda.synchronize()
        print(n, 'outputs per channel: %.1f us per call' % ((time.perf_counter()-t0)/K*1e6))
