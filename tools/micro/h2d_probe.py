import time, torch
from torch.utils.data import DataLoader, Dataset
class D(Dataset):
    def __len__(self): return 16
    def __getitem__(self, i): return torch.randn(3,256,256), torch.zeros(1,256,256), torch.randn(3,256,256)
dev=torch.device('cuda',0)
torch.zeros(1,device=dev); torch.cuda.synchronize()
def t(f,n=1):
    torch.cuda.synchronize(); t0=time.perf_counter(); r=f(); torch.cuda.synchronize(); return (time.perf_counter()-t0)*1e3
x=torch.randn(1,3,256,256)
print('plain cpu tensor .to:', [round(t(lambda: x.to(dev)),3) for _ in range(4)])
for nw in (0,2):
    dl=DataLoader(D(),batch_size=1,num_workers=nw)
    ts=[]
    for b in dl:
        ts.append(round(t(lambda: [u.to(dev) for u in b]),3))
    print('loader workers',nw,ts[:8])
stage=torch.empty(1,3,256,256).pin_memory()
dl=DataLoader(D(),batch_size=1,num_workers=2)
ts=[]
for b in dl:
    ts.append(round(t(lambda: (stage.copy_(b[0]), stage.to(dev))),3))
print('via pinned stage (one tensor)',ts[:8])
dl=DataLoader(D(),batch_size=1,num_workers=2)
ts=[]
for b in dl:
    ts.append(round(t(lambda: [u.clone().to(dev) for u in b]),3))
print('clone() first',ts[:8])
