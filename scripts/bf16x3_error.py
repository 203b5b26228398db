"""Numerical experiment behind csrc/mlpb.hip: error of 128-term dot products with
three-part bf16 operands and 3 / 6 / 8 / 9 of the nine partial products (exact and
fp32 accumulation per 32-term block), against an fp32 FMA chain; truth = fp64.
CPU only (numpy)."""
import numpy as np
rng=np.random.default_rng(0)
def trunc_bf16(x):
    u=x.astype(np.float32).view(np.uint32)&np.uint32(0xFFFF0000)
    return u.view(np.float32)
def rn_bf16(x):
    u=x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r=((u+0x7FFF+((u>>16)&1))&0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)
def split(x,rnd):
    f=rn_bf16 if rnd else trunc_bf16
    b0=f(x); r1=(x-b0).astype(np.float32); b1=f(r1); r2=(r1-b1).astype(np.float32); b2=f(r2)
    return b0,b1,b2,(r2-b2)
K=128; N=20000
a=rng.standard_normal((N,K)).astype(np.float32); b=rng.standard_normal((N,K)).astype(np.float32)
truth=(a.astype(np.float64)*b.astype(np.float64)).sum(1)
scale=np.sqrt((a.astype(np.float64)**2).sum(1)*(b.astype(np.float64)**2).sum(1))
# fp32 sequential accumulate
acc=np.zeros(N,np.float32)
for k in range(K): acc=(acc+a[:,k]*b[:,k]).astype(np.float32)   # product rounded too (no fma) - pessimistic
acc2=np.zeros(N,np.float32)
for k in range(K): acc2=(acc2.astype(np.float64)+a[:,k].astype(np.float64)*b[:,k]).astype(np.float32) # fma
print("fp32 mul+add rms err/scale", np.sqrt(np.mean(((acc-truth)/scale)**2)))
print("fp32 fma     rms err/scale", np.sqrt(np.mean(((acc2-truth)/scale)**2)))
for rnd in (0,1):
    A=split(a,rnd); B=split(b,rnd)
    print("rnd",rnd,"residual a max rel", np.abs(A[3]).max())
    for name,terms in (("3",[(0,0),(0,1),(1,0)]),("6",[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]),("8",[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0),(1,2),(2,1)]),("9",[(i,j) for i in range(3) for j in range(3)])):
        s=np.zeros(N)
        for i,j in terms: s+= (A[i].astype(np.float64)*B[j].astype(np.float64)).sum(1)
        print("  bf16x%s exact-accumulate rms err/scale"%name, np.sqrt(np.mean(((s-truth)/scale)**2)))
        # fp32 accumulate per k-block of 32 (mfma) : emulate acc in fp32 after each 32-block of all terms
        acc=np.zeros(N,np.float32)
        for kb in range(0,K,32):
            # small terms first into separate accumulators like M/X ? here: one accumulator, order small->large
            for i,j in sorted(terms,key=lambda t:-(t[0]+t[1])):
                acc=(acc.astype(np.float64)+(A[i][:,kb:kb+32].astype(np.float64)*B[j][:,kb:kb+32]).sum(1)).astype(np.float32)
        print("  bf16x%s fp32-accumulate  rms err/scale"%name, np.sqrt(np.mean(((acc-truth)/scale)**2)))
