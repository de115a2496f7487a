#!/usr/bin/env python3
"""Tuning experiment: per-phase cycle counts inside one workgroup of the field kernel (s_memtime marks).
Build the instrumented library first (marks are compiled out of the product build):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -pragma-unroll-threshold=1000000 \
        -DMNRF_EXP_MARKS -c mirror_nerf_amd/csrc/mnrf_field.hip -o /tmp/f.o
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/f.o mirror_nerf_amd/csrc/mnrf_render.o mirror_nerf_amd/csrc/mnrf_dw.o \
        mirror_nerf_amd/csrc/mnrf_tcnn.o -o /tmp/libmnrf_marks.so
  MNRF_LIB=/tmp/libmnrf_marks.so MNRF_FIELD_VARIANT=s2 python scripts/exp_phase_marks.py
(-DMNRF_EXP_NO_DMA / -DMNRF_EXP_NO_READ build the "no barrier+DMA" / "no LDS read" timing variants the
same way; their results are garbage by construction.)"""
import sys, torch
sys.path.insert(0,'.')
import bench, mirror_nerf_amd as M
from mirror_nerf_amd import _lib
from mirror_nerf_amd.weights import packed_of
from oracle import mirror_nerf_oracle as O
dev=torch.device("cuda",0)
models,sds,emb=bench.build_models(dev)
rays=torch.from_numpy(O.synthetic_rays(800,800)[300*800:300*800+32768]).to(dev)
S=192
z=torch.sort(torch.rand(32768,S,device=dev)*7+0.05,1)[0].contiguous()
dir_emb=emb["dir"](rays[:,3:6].contiguous())
packed=packed_of(models["fine"])
B=32768*S
f=lambda *s: torch.empty(*s,device=dev)
sig,rgb,pn,mir=f(B),f(B,3),f(B,3),f(B)
dbg=torch.zeros(64,dtype=torch.int64,device=dev)
p=_lib.ptr
for it in range(3):
    _lib.check(_lib.lib().mnrf_field_forward(p(packed),0,B,None,3,p(rays),p(z),S,p(dir_emb),27,p(sig),p(rgb),p(pn),p(mir),p(dbg),None,_lib.stream()),"f")
torch.cuda.synchronize()
t=dbg.cpu().tolist()[:13]
names=["start","bias+open","positions","encoding","L1","L2-4","L5","L6-8","geo+SIG","NRM","MIR","FIN","DIR+RGB"]
print("total", t[12]-t[0])
for i in range(1,13): print(f"{names[i]:10s} {t[i]-t[i-1]:8d}")
