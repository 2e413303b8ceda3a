import sys, os
sys.path.insert(0, '/root/repo')
import torch
from pwv_amd import engine
from pwv_amd.hparam import hparam as hp
from pwv_amd.models import IAFVocoder
from pwv_amd.variables import VariableStore
case, utts = sys.argv[1], int(sys.argv[2])
hp.set_hparam_yaml(case)
dev = torch.device('cuda', 0)
length = hp.generate.length
store = VariableStore(device=dev, seed=2)
model = IAFVocoder(batch_size=utts, length=length, store=store)
model.noise_seed = 1
g = torch.Generator().manual_seed(1000)
mel = (torch.rand((utts, 1 + length // 80, 80), generator=g) * 2 - 1).to(dev)
model(None, mel)
torch.manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
for name in list(store.vars):
    if store.vars[name].dim() == 1:
        store.vars[name].normal_(0, 0.1)
store.version += 1
engine.clear_plan_cache()
orig = engine.run_nets
def spy(nets, x, cond, precision=None, max_workgroups=0):
    out = orig(nets, x, cond, precision, max_workgroups)
    plans = [engine.get_plan(n, 'frames', engine._lib.PREC_F16X3) for n in nets]
    print('flow input max|x| %.4g  (x_limit %.4g)  net outputs max %s' % (float(x.abs().max()), min(p.x_limit for p in plans), [float(o.abs().max()) for o in out]))
    return out
engine.run_nets = spy
import pwv_amd.modules as M
y = model(None, mel)
torch.cuda.synchronize()
print('output max|y| %.4g, finite %s, flag %s' % (float(y.abs().max()), bool(torch.isfinite(y).all()), engine.range_flag_raised()))
engine.run_nets = orig
from pwv_amd.graph import GraphedVocoder
gv = GraphedVocoder(model)
print('after capture: flag', engine.range_flag_raised())
for k in range(6):
    y = gv(mel)
    torch.cuda.synchronize()
    print('replay %d: max|y| %.4g flag %s' % (k, float(y.abs().max()), engine.range_flag_raised()))
