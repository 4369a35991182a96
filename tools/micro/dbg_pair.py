import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from timeband_amd import ops
from timeband_amd.utils.initiate import seeding
from timeband_amd.utils.lstm_layer import LSTMDiscriminator
dev=torch.device('cuda',0)
seeding(31)
netD=LSTMDiscriminator(32,256,1,dev).to(dev)
g=torch.Generator(device='cpu').manual_seed(7)
y=(torch.rand(1024,128,32,generator=g)*2-1).to(dev)
with torch.no_grad():
    l=netD.lstm
    a=ops.lstm_layer(y, l.weight_ih_l0,l.weight_hh_l0,l.bias_ih_l0,l.bias_hh_l0)
    b=torch.cat([ops.lstm_layer(y[:512], l.weight_ih_l0,l.weight_hh_l0,l.bias_ih_l0,l.bias_hh_l0), ops.lstm_layer(y[512:], l.weight_ih_l0,l.weight_hh_l0,l.bias_ih_l0,l.bias_hh_l0)],0)
    a2=ops.lstm_layer(y, l.weight_ih_l0,l.weight_hh_l0,l.bias_ih_l0,l.bias_hh_l0)
torch.cuda.synchronize(); ops.check_errors()
d=(a-b).abs()
print("max diff", d.max().item(), "repeat diff", (a-a2).abs().max().item())
bad=(d>0).nonzero()
print("n bad", bad.shape[0], "of", d.numel())
if bad.shape[0]:
    rows=torch.unique(bad[:,0]); ts=torch.unique(bad[:,1]); us=torch.unique(bad[:,2])
    print("rows", rows[:40].tolist(), len(rows)); print("t", ts[:20].tolist(), len(ts)); print("units", us[:40].tolist(), len(us))
    r0=bad[0]; print("first", r0.tolist(), a[tuple(r0)].item(), b[tuple(r0)].item())
