#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/pytest.txt
tail -3 gpurun_out/pytest.txt
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0,'tools'); sys.path.insert(0,'tests')
import gpu_tune, datagen
from deepblast_amd._engine import get_engine
eng=get_engine()
B=1024; lens=datagen.lengths(9,B,64,512); N,M=512,512
th,A=datagen.theta_A(9,64,N,M); th=torch.from_numpy(np.tile(th,(16,1,1))).cuda(); A=torch.from_numpy(np.tile(A,(16,1,1))).cuda()
et=torch.ones(B,device='cuda')
def run(tag, ln):
    lnt=torch.from_numpy(ln).cuda(); st={}
    def f(): st['v'],st['q']=eng.forward(th,A,0,lnt)
    def b(): st['e']=eng.backward(et,st['q'],(B,N,M),0,lnt)
    f(); tf=gpu_tune.timeit(f,5); tb=gpu_tune.timeit(b,5)
    w=int((ln[:,0].astype(np.int64)*ln[:,1]).sum())
    print(f"{tag}: fwd={tf:.1f} bwd={tb:.1f} us  {2*w/(tf+tb)*1e6:.3e} true cell-updates/s", flush=True)
run("B=1024 U[64,512] lens (library orders longest-first)", lens)
PY
