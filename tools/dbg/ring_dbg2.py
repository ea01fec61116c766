import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cadre_amd import hip
from cadre_amd.encoder import _ring_w
def run(F,H,W,Cin,N,tap):
    td=torch.bfloat16
    # x[p][c] = p + c/256 (distinct, exact in fp32; bf16 rounding ok for small ints): use integers: x = (p % 64)*2 + ... keep small
    M=F*H*W
    p=torch.arange(M).reshape(F,H,W,1).float(); c=torch.arange(Cin).reshape(1,1,1,Cin).float()
    x=((p % 97) + c*0.0).to(td).cuda() if False else None
    xp=(p % 101).expand(F,H,W,Cin).contiguous()        # value = position id mod 101
    xc=c.expand(F,H,W,Cin).contiguous()                # value = channel id
    for name,xx in (("pos",xp),("chan",xc)):
        x=xx.to(td).cuda()
        w=torch.zeros(N,Cin,3,3)
        for n in range(min(N,Cin)): w[n,n,tap//3,tap%3]=1.0
        wr=_ring_w(w,64).to(td).cuda()
        out=torch.full((F,H,W,N),-1.0,device="cuda",dtype=td)
        hip.conv3x3_ring(x,wr,None,None,None,out,F,H,W,Cin,N,0)
        torch.cuda.synchronize()
        o=out.float().cpu().reshape(M,N)
        ref=torch.nn.functional.conv2d(xx.permute(0,3,1,2),w,padding=1).permute(0,2,3,1).reshape(M,N)
        print(name,"tap",tap,"max err",float((o-ref).abs().max()))
        for pos in (0,1,17,18,40,100,200,255):
            print("  pos",pos,"got",[int(v) for v in o[pos,:16]],"...",[int(v) for v in o[pos,32:40]],"want",[int(v) for v in ref[pos,:4]],[int(v) for v in ref[pos,32:36]])
run(1,16,16,64,64,4)
run(1,16,16,64,64,0)
