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
