import torch
torch.manual_seed(0)
dev="cuda"
for (M,K,N) in [(1280,1024,4096),(1280,151936,1024),(151936,1280,1024),(1280,1024,151936)]:
    a=torch.randn(M,K,device=dev); b=torch.randn(K,N,device=dev)
    ref=(a.double()@b.double())
    for flag in (False, True):
        torch.backends.cuda.matmul.allow_tf32=flag
        c=a@b
        print((M,K,N),"allow_tf32",flag,"rel err", float((c.double()-ref).norm()/ref.norm()))
print(torch.get_float32_matmul_precision())
