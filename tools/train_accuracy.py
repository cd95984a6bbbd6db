import sys
sys.path.insert(0,'/root/repo/pdp-solver_amd'); sys.path.insert(0,'/root/repo/tests')
import torch, torch.nn.functional as F
from pdp.nn import train_ops as T
from pdp import native
DEV='cuda:0'
def leaf(*shape, scale=0.3, seed=0):
    g = torch.Generator(device='cpu'); g.manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).requires_grad_(True)
def rel(a, r): return float((a.double() - r).abs().max() / r.abs().max().clamp(min=1e-30))
for R,K,N,act in [(1000,33,100,'logsigmoid'),(4097,51,128,'relu'),(5000,151,100,'none'),(8200,129,100,'logsigmoid'),(4200,100,50,'logsigmoid'),(6000,129,384,'none'),(100000,129,100,'logsigmoid')]:
    x,w,b = leaf(R,K,seed=1), leaf(N,K,seed=2), leaf(N,seed=3)
    fn = {'logsigmoid': F.logsigmoid, 'relu': torch.relu, 'none': lambda z: z}[act]
    g = torch.randn(R,N,device=DEV)
    y = T.LinearAct.apply(x,w,b,act); y.backward(g)
    got=[y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    xd,wd,bd=[t.detach().double().requires_grad_(True) for t in (x,w,b)]
    yr = fn(F.linear(xd,wd,bd)); yr.backward(g.double())
    # torch fp32 for comparison
    x2,w2,b2=[t.detach().clone().requires_grad_(True) for t in (x,w,b)]
    y2 = fn(F.linear(x2,w2,b2)); y2.backward(g)
    print('linear',R,K,N,act,'ours vs f64:', ['%.1e'%rel(a,r) for a,r in zip(got,(yr.detach(),xd.grad,wd.grad,bd.grad))], ' torch f32 vs f64:', ['%.1e'%rel(a,r) for a,r in zip((y2.detach(),x2.grad,w2.grad,b2.grad),(yr.detach(),xd.grad,wd.grad,bd.grad))])
for R,Ks in ((64*41+7,128),(64*9+5,3),(50000,128)):
    cell=torch.nn.GRUCell(Ks+1,128).to(DEV)
    state,h=leaf(R,Ks,seed=21),leaf(R,128,seed=22)
    sign=torch.sign(torch.randn(R,1,device=DEV))
    packed=native.GruWeights(cell.weight_ih.data,cell.weight_hh.data,cell.bias_ih.data,cell.bias_hh.data)
    g=torch.randn(R,128,device=DEV)
    hn=T.GruCellS.apply(state,sign,h,cell.weight_ih,cell.weight_hh,cell.bias_ih,cell.bias_hh,packed); hn.backward(g)
    got=[hn.detach().clone(),state.grad.clone(),h.grad.clone()]+[p.grad.clone() for p in cell.parameters()]
    import copy
    cd=copy.deepcopy(cell).double(); sd,hd=state.detach().double().requires_grad_(True),h.detach().double().requires_grad_(True)
    hr=cd(torch.cat((sd,sign.double()),1),hd); hr.backward(g.double())
    ref=[hr.detach(),sd.grad,hd.grad]+[p.grad for p in cd.parameters()]
    c2=copy.deepcopy(cell); [setattr(p,'grad',None) for p in c2.parameters()]
    s2,h2=state.detach().clone().requires_grad_(True),h.detach().clone().requires_grad_(True)
    r2=c2(torch.cat((s2,sign),1),h2); r2.backward(g)
    t32=[r2.detach(),s2.grad,h2.grad]+[p.grad for p in c2.parameters()]
    print('gru',R,Ks,'ours vs f64:',['%.1e'%rel(a,r) for a,r in zip(got,ref)],' torch f32 vs f64:',['%.1e'%rel(a,r) for a,r in zip(t32,ref)])
