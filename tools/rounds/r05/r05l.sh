#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
export DECNET_HIP_LIB=$R/tools/ubench/libdecnet_dev_r05midk.so
for k in 1 0; do
  for d in 1.0 0.6 0.5 0.4 0.3 0.25 0.2 0.1 0.05; do
    echo -n "midk=$k " >> $O/times.txt
    DECNET_SPAMAT_MIDK=$k python3 tools/bench_spamat.py --stage 3 --density $d --iters 30 2>/dev/null >> $O/times.txt
  done
done
for d in 0.5 0.3 0.1; do echo -n "midk=1 bits " >> $O/times.txt; python3 tools/bench_spamat.py --stage 3 --density $d --iters 30 --bits 2>/dev/null >> $O/times.txt; done
python3 - <<'PY' >> $O/times.txt 2>&1
import torch, decnet_amd, oracle, numpy as np
dev=torch.device('cuda:0')
g=torch.Generator().manual_seed(9)
for (C,H,W,D,p) in ((8,6,972,216,0.3),(8,6,972,216,0.5),(8,4,1000,216,0.4),(8,3,700,216,0.45),(7,3,600,100,0.6),(8,5,972,216,0.58)):
    L=torch.randn(2,C,H,W,generator=g); R=torch.randn(2,C,H,W,generator=g)
    rm=(torch.rand(2,H,W,generator=g)<p).float(); tm=(torch.rand(2,H,W,generator=g)<p).float()
    o,v,s,m=decnet_amd.spamatvar_forward(L.to(dev),R.to(dev),rm.to(dev),tm.to(dev),D)
    oo,ss,mm=oracle.spamat_forward(L,R,rm,tm,D); vv,_,_=oracle.spavar_forward(L,R,rm,tm,oo,D)
    print(C,W,D,p,'disp max err %.2e'%np.abs(o.cpu().numpy()-oo).max(),'var rel %.2e'%(np.abs(v.cpu().numpy()-vv)/(np.abs(vv)+1e-2)).max(),'mx rel %.2e'%(np.abs(m.cpu().numpy()-mm)/(np.abs(mm)+1e-9)).max(),'S rel %.2e'%(np.abs(s.cpu().numpy()-ss)/(np.abs(ss)+1e-9)).max())
PY
cat $O/times.txt | sed 's/algorithmic //; s/stage 3 fused C=8 H=540 W=972 D=216 B=8 //'
