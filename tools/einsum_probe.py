import torch
torch.manual_seed(0)
print("threads", torch.get_num_threads(), torch.__config__.parallel_info().split("\n")[0])
import subprocess; print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Flags' | cut -c1-200", shell=True, capture_output=True, text=True).stdout)
for N in (40, 1000, 4096):
    M = torch.randn(2,1,4,4); x = torch.randn(2,N,4); x[...,3]=1
    ref = torch.einsum('...ij,...j->...i', M, x)
    Md, xd = M.double(), x.double()
    def f32(t): return t.float().double()
    def chain(fma, order=range(4)):
        out=None
        for j in order:
            prod = Md[..., :, j] * xd[..., j:j+1]
            if out is None: out=f32(prod)
            else: out = f32(out+prod) if fma else f32(out+f32(prod))
        return out.float()
    print(N, 'fma chain', (chain(True)==ref).float().mean().item(), 'plain', (chain(False)==ref).float().mean().item(),
          'fma rev', (chain(True, [3,2,1,0])==ref).float().mean().item())
    # 3x3
    K = torch.randn(2,1,3,3); y = torch.randn(2,N,3)
    r3 = torch.einsum('...ij,...j->...i', K, y)
    Kd, yd = K.double(), y.double()
    out=None
    for j in range(3):
        prod = Kd[..., :, j]*yd[..., j:j+1]
        out = f32(prod) if out is None else f32(out+prod)
    print(N, '3x3 fma chain', (out.float()==r3).float().mean().item())
