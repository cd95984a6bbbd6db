import os, sys
sys.path.insert(0, '/root/repo/pdp-solver_amd')
import torch, ctypes as C
from pdp import native
from pdp.nn import train_ops as T
dev = torch.device('cuda:0')
torch.manual_seed(0)
R = 64 * 37 + 13
cell = torch.nn.GRUCell(129, 128).to(dev)
state = torch.randn(R, 128, device=dev) * 0.5; sign = torch.sign(torch.randn(R, 1, device=dev)); h = torch.randn(R, 128, device=dev) * 0.5
x = torch.cat((state, sign), 1)
packed = native.GruWeights(cell.weight_ih.data, cell.weight_hh.data, cell.bias_ih.data, cell.bias_hh.data)
class Ctx: 
    def save_for_backward(self, *a): self.saved = a
c1, c2 = Ctx(), Ctx()
with torch.no_grad():
    h1 = T.GruCell.forward(c1, x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, packed, state, sign)
    os.environ['PDP_TRAIN_GRU'] = 'plain'
    h2 = T.GruCell.forward(c2, x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    ref = cell(x, h)
print('hnew fused-plain', float((h1 - h2).abs().max()), 'fused-torch', float((h1 - ref).abs().max()), 'plain-torch', float((h2 - ref).abs().max()))
s1, s2 = c1.saved[4], c2.saved[4]
for g, nm in enumerate(('r', 'z', 'n', 'ghn')):
    d = (s1[:, g * 128:(g + 1) * 128] - s2[:, g * 128:(g + 1) * 128]).abs()
    print(nm, float(d.max()), int(d.argmax()) // 128, float(s2[:, g * 128:(g + 1) * 128].abs().max()))
# where do mismatching values come from?
bad = ((s1[:2368, :128] - s2[:2368, :128]).abs() > 1e-4)
print('bad fraction r', float(bad.float().mean()), 'rows with bad', int(bad.any(1).sum()), 'cols with bad', int(bad.any(0).sum()))
rows = bad.any(1).nonzero().reshape(-1)[:10].tolist(); print('first bad rows', rows, 'mod 64', [r % 64 for r in rows])
r0 = rows[0]; cols = bad[r0].nonzero().reshape(-1)[:5].tolist(); print('bad cols in row', r0, cols)
v = float(s1[r0, cols[0]])
hit = ((s2 - v).abs() < 1e-6).nonzero()[:8].tolist(); print('value', v, 'found in plain saved at', hit)
