import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cadre_amd import hip
from cadre_amd.encoder import _ring_w
F,H,W,Cin,N=1,16,16,64,64
td=torch.bfloat16
M=F*H*W
p=torch.arange(M).reshape(F,H,W,1).float(); c=torch.arange(Cin).reshape(1,1,1,Cin).float()
xx=((p%4)*64+c).contiguous()
x=xx.to(td).cuda()
for tap in (4,):
    w=torch.zeros(N,Cin,3,3)
    for n in range(N): w[n,n,tap//3,tap%3]=1.0
    wr=_ring_w(w,64).to(td).cuda()
    out=torch.full((F,H,W,N),-1.0,device="cuda",dtype=torch.float32)
    hip.conv3x3_ring(x,wr,None,None,None,out,F,H,W,Cin,N,0)
    torch.cuda.synchronize()
    o=out.cpu().reshape(M,N)
    for pos in (0,1,2,3,33,34,100):
        print("pos",pos,[int(v) for v in o[pos]])
# weights test: w[n][c] = 1 if c == (n*7)%64 : permutation; x[p][c]=c
perm=[(n*7+3)%64 for n in range(N)]
w=torch.zeros(N,Cin,3,3)
for n in range(N): w[n,perm[n],1,1]=1.0
wr=_ring_w(w,64).to(td).cuda()
x=c.expand(F,H,W,Cin).contiguous().to(td).cuda()
out=torch.full((F,H,W,N),-1.0,device="cuda",dtype=torch.float32)
hip.conv3x3_ring(x,wr,None,None,None,out,F,H,W,Cin,N,0)
torch.cuda.synchronize()
o=out.cpu().reshape(M,N)
print("perm want",perm)
print("perm got ",[int(v) for v in o[5]])
# single row weights: only output channel n0 has weights: w[n0][c]=1 for c==5 -> out[:, n0] = x[:,5]=5 others 0
for n0 in (0,1,9,31,33):
    w=torch.zeros(N,Cin,3,3); w[n0,5,1,1]=1.0
    wr=_ring_w(w,64).to(td).cuda()
    out=torch.full((F,H,W,N),-1.0,device="cuda",dtype=torch.float32)
    hip.conv3x3_ring(x,wr,None,None,None,out,F,H,W,Cin,N,0)
    torch.cuda.synchronize()
    o=out.cpu().reshape(M,N)
    print("only row",n0,"->",[int(v) for v in o[7]])
