import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, mon_amd
from tests.helpers import load_golden, zero_dce_goldens
from oracle import zero_dce as o
path=[p for p in zero_dce_goldens() if '37x131' in p][0]
sd,g=load_golden(path)
x=torch.from_numpy(g['image'])
ref=o.forward(sd,x,return_features=True)
for dtype in ('f16',):
    m=mon_amd.MODELS.build(name='zero_dce_re',weights=dict(sd),compute_dtype=dtype).to('cuda')
    out=m.forward({'image':x.cuda()})
    e=(out['enhanced'].cpu()-ref['enhanced']).abs().amax(dim=(0,1))
    print('enh err max',float(e.max()))
    bad=(e>1e-3).nonzero()
    print('bad count',len(bad),'rows',sorted(set(bad[:,0].tolist())),'cols',sorted(set(bad[:,1].tolist()))[:40])
    for k in (1,2,3,5):
        f=m.feature(k).cpu(); r=ref['_features'][k-1]
        ee=(f-r).abs().amax(dim=(0,1)); b=(ee>0.02*float(r.abs().max())).nonzero()
        print('x',k,'max',float(ee.max()),'bad',len(b),'rows',sorted(set(b[:,0].tolist()))[:20],'cols',sorted(set(b[:,1].tolist()))[:20])
