"""Brute-force search of the XOR swizzles of the LDS images of csrc/mlpb.hip with the
bank model of scripts/lds_banks.py (first version of the kernel: H1 / dY images with
16-byte row reads + transpose reads, X image)."""
import itertools, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from lds_banks import G128, HALF, W128, W64, worst
def lane(l): return l&15, l>>4
def lane_tr(l): return l>>4, (l>>2)&3, l&3
def mk(masks):
    def swz(r):
        v=0
        for b,m in enumerate(masks):
            if (r>>b)&1: v^=m
        return v
    return swz
def eval_T(P, swz):
    A=lambda row,col: row*P+(col^swz(row))
    tot=[]
    # b128 write of own slice
    tot.append(max(worst([A(16*nb+lane(l)[0], 64*w+16*lane(l)[1]) for l in range(64)], W128,16,32) for nb in range(4) for w in range(4)))
    # b128 read B frags
    tot.append(max(worst([A(16*nb+lane(l)[0], 64*kb+16*lane(l)[1]) for l in range(64)], G128,16,64) for nb in range(4) for kb in range(4)))
    # tr64 reads
    tot.append(max(worst([A(32*kb+16*s+4*lane_tr(l)[0]+lane_tr(l)[1], 32*ub+8*lane_tr(l)[2]) for l in range(64)], HALF,8,64) for kb in range(2) for s in range(2) for ub in range(8)))
    return tot
def eval_X(P, swz):
    A=lambda row,col: row*P+(col^swz(row))
    tot=[]
    tot.append(max(worst([A(16*w+(l>>2), 32*(l&3)+16*h) for l in range(64)], W128,16,32) for w in range(4) for h in range(2)))
    tot.append(max(worst([A(16*nb+lane(l)[0], 64*kb+16*lane(l)[1]) for l in range(64)], G128,16,64) for nb in range(4) for kb in range(2)))
    tot.append(max(worst([A(32*kb+16*s+4*lane_tr(l)[0]+lane_tr(l)[1], 32*fb+8*lane_tr(l)[2]) for l in range(64)], HALF,8,64) for kb in range(2) for s in range(2) for fb in range(4)))
    return tot
best=None
opts=[0,16,32,48,64,80,96,112,128,144,160,176,192,208,224,240]
for masks in itertools.product(opts, repeat=4):
    t=eval_T(256, mk(masks))
    k=(max(t),sum(t))
    if best is None or k<best[0]:
        best=(k,masks,t); print("T",best)
        if k==(1,3): break
bestx=None
optx=[0,16,32,48,64,80,96,112]
for masks in itertools.product(optx, repeat=4):
    t=eval_X(128, mk(masks))
    k=(max(t),sum(t))
    if bestx is None or k<bestx[0]:
        bestx=(k,masks,t); print("X",bestx)
        if k==(1,3): break
