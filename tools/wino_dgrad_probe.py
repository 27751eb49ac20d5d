import sys, os
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from jmodt_amd.ops.fusion import conv3x3_wino_bias_relu, pack_wino_weight
torch.manual_seed(0)
dev="cuda:0"
def rel(a,b): return ((a.double()-b.double()).abs().max()/b.double().abs().max()).item()
for (H,W) in [(24,40),(32,48)]:
  for cin,cout in [(64,128),(128,256)]:
    B=2
    x=torch.randn(B,cin,H,W,device=dev).contiguous(memory_format=torch.channels_last)
    w=(torch.randn(cout,cin,3,3,device=dev)*0.1).contiguous(memory_format=torch.channels_last)
    b=torch.randn(cout,device=dev)*0.1
    g=torch.randn(B,cout,H,W,device=dev).contiguous(memory_format=torch.channels_last)
    x0=x.clone(); w0=w.clone()
    y=conv3x3_wino_bias_relu(x, pack_wino_weight(w), b, cout)
    torch.cuda.synchronize()
    print(H,W,cin,cout,"x intact after fwd", torch.equal(x,x0), "w intact", torch.equal(w,w0))
    yref=F.relu(F.conv2d(x.double(),w.double(),b.double(),padding=1))
    print("   fwd rel err", rel(y,yref))
    dpre=torch.ops.aten.threshold_backward(g,y,0)
    d0=dpre.clone()
    gx,dw,db=torch.ops.aten.convolution_backward(dpre,x,w,[cout],[1,1],[1,1],[1,1],False,[0,0],1,[False,True,True])
    x2=x.double().requires_grad_(); w2=w.double().requires_grad_(); b2=b.double().requires_grad_()
    r=F.relu(F.conv2d(x2,w2,b2,padding=1)); r.backward(g.double())
    print("   dW (mask F,T,T) rel err", rel(dw,w2.grad), "db", rel(db,b2.grad))
    gx,dw2,db2=torch.ops.aten.convolution_backward(dpre,x,w,[cout],[1,1],[1,1],[1,1],False,[0,0],1,[True,True,True])
    print("   dW (mask T,T,T) rel err", rel(dw2,w2.grad), "dX miopen", rel(gx,x2.grad))
    dx=conv3x3_wino_bias_relu(dpre, pack_wino_weight(w.flip(2,3).transpose(0,1)), None, cin, relu=False)
    torch.cuda.synchronize()
    print("   dX wino rel err", rel(dx,x2.grad), "dpre intact", torch.equal(dpre,d0))
print("---- order: wino dX first, then MIOpen dW (what _Conv3x3BiasRelu.backward does)")
for cin,cout in [(64,128),(128,256)]:
    B,H,W=2,24,40
    x=torch.randn(B,cin,H,W,device=dev).contiguous(memory_format=torch.channels_last)
    w=(torch.randn(cout,cin,3,3,device=dev)*0.1).contiguous(memory_format=torch.channels_last)
    g=torch.randn(B,cout,H,W,device=dev).contiguous(memory_format=torch.channels_last)
    x0=x.clone(); w0=w.clone(); g0=g.clone()
    dx=conv3x3_wino_bias_relu(g, pack_wino_weight(w.flip(2,3).transpose(0,1)), None, cin, relu=False)
    torch.cuda.synchronize()
    print(cin,cout,"after wino dX: x intact", torch.equal(x,x0), "w intact", torch.equal(w,w0), "g intact", torch.equal(g,g0))
    gx,dw,db=torch.ops.aten.convolution_backward(g,x,w,[cout],[1,1],[1,1],[1,1],False,[0,0],1,[False,True,True])
    gx2,dw2,db2=torch.ops.aten.convolution_backward(g0,x0,w0,[cout],[1,1],[1,1],[1,1],False,[0,0],1,[False,True,True])
    print("   dW after wino vs clean operands", rel(dw,dw2))
    w2=w0.double().requires_grad_(); F.conv2d(x0.double(),w2,None,padding=1).backward(g0.double())
    print("   dW clean vs float64", rel(dw2,w2.grad), "| dW via Function path", rel(dw,w2.grad))
    # the flipped weight as a channels-last view? the test passes a channels-last w: pack_wino_weight(w.flip.transpose) on cl memory
    wf=w.flip(2,3).transpose(0,1)
    print("   flipped/transposed strides", wf.stride(), "contiguous copy equal:", torch.equal(wf.contiguous(), wf.clone(memory_format=torch.contiguous_format)))
