#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_s12cb2.so
for st in 2 1; do for d in 1.0 0.5; do python3 tools/bench_spamat.py --stage $st --density $d --iters 100 2>/dev/null >> $O/times.txt; done; done
for st in 2 1; do DECNET_SPAMAT_DENSE=fp32 python3 tools/bench_spamat.py --stage $st --density 1.0 --iters 100 2>/dev/null | sed 's/^/fp32: /' >> $O/times.txt; done
python3 - <<'PY' >> $O/times.txt 2>&1
import torch, decnet_amd, oracle, numpy as np
dev=torch.device('cuda:0')
g=torch.Generator().manual_seed(5)
for (C,H,W,D) in ((24,7,324,72),(72,5,108,24),(24,3,100,72),(72,3,50,24),(20,4,200,40)):
    L=torch.relu(torch.randn(2,C,H,W,generator=g)); R=torch.relu(torch.randn(2,C,H,W,generator=g))
    rm=(torch.rand(2,H,W,generator=g)<0.9).float(); tm=(torch.rand(2,H,W,generator=g)<0.9).float()
    o,v,s,m=decnet_amd.spamatvar_forward(L.to(dev),R.to(dev),rm.to(dev),tm.to(dev),D)
    oo,ss,mm=oracle.spamat_forward(L,R,rm,tm,D); vv,_,_=oracle.spavar_forward(L,R,rm,tm,oo,D)
    print(C,W,D,'disp max err %.2e'%np.abs(o.cpu().numpy()-oo).max(),'var rel %.2e'%(np.abs(v.cpu().numpy()-vv)/(np.abs(vv)+1e-3)).max(),'mx rel %.2e'%(np.abs(m.cpu().numpy()-mm)/(np.abs(mm)+1e-9)).max())
PY
cat $O/times.txt | sed 's/algorithmic //'
