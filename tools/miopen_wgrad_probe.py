import torch, torch.nn.functional as F
torch.manual_seed(0)
dev="cuda:0"
for cin,cout in [(64,128),(128,256),(256,512)]:
    for cl in (False, True):
        B,H,W=2,24,40
        x=torch.randn(B,cin,H,W,device=dev); w=(torch.randn(cout,cin,3,3,device=dev)*0.1); g=torch.randn(B,cout,H,W,device=dev)
        if cl:
            x=x.contiguous(memory_format=torch.channels_last); w=w.contiguous(memory_format=torch.channels_last); g=g.contiguous(memory_format=torch.channels_last)
        gx,dw,db=torch.ops.aten.convolution_backward(g,x,w,[cout],[1,1],[1,1],[1,1],False,[0,0],1,[True,True,True])
        x2=x.double().requires_grad_(); w2=w.double().requires_grad_()
        y=F.conv2d(x2,w2,None,padding=1); y.backward(g.double())
        e=(dw.double()-w2.grad).abs().max().item(); s=w2.grad.abs().max().item()
        ex=(gx.double()-x2.grad).abs().max().item(); sx=x2.grad.abs().max().item()
        print(cin,cout,"channels_last" if cl else "nchw", f"dW err {e:.3e} / {s:.1f} = {e/s:.2e};  dX err {ex:.3e} / {sx:.1f} = {ex/sx:.2e}")
