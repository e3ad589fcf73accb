"""Which summation order of the aero MLPs' Linear layers lands closest to the authors' CUDA recording (renders/result/*.npy, 426 steps)?
Build container only (imports /root/reference; nothing of it is copied): the reference's F16Dynamics.nlplant replays the recorded controls with
the MLP forward replaced by explicit fp32 orders.  Printed: (worst rel. error up to step 400, up to step 426) against the recording, SURVEY floors.
Round 6 result (profiles/r06h_mlp_order_probe.log): the reference's own ATen-CPU 2.4e-5; fma chain from ZERO with the bias added LAST 3.9e-5;
the shipped spec's order (bias first, then the fma chain) 2.6e-4 here (1.76e-4 with its two-chain output layer); per-layer exact 2.2e-4.
    python tools/microbench/mlp_order_probe.py"""
import sys, os, io, contextlib
sys.path[:0] = ['/root/repo/tools/oracle_shims', '/root/reference', '/root/reference/envs']
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
torch.set_num_threads(1)
from envs.models.F16.F16_dynamics import F16Dynamics
with contextlib.redirect_stdout(io.StringIO()):
    dyn = F16Dynamics('cpu')
d='/root/reference/renders/result'
cols=['npos','epos','altitude','roll','pitch','yaw','vt','alpha','beta','G','T','el','ail','rud']
arr=np.stack([np.load(os.path.join(d,c+'.npy')).reshape(-1)[:427] for c in cols],1).astype(np.float32)
floors=np.array([100,100,100,.1,.1,.1,10,.1,.1],np.float32)
mlp_cls=type(dyn.hifi_F16.Cx_model)
o_fwd=mlp_cls.forward
def replay():
    s=torch.zeros(1,12); s[0,2]=float(arr[0,2]); s[0,6]=float(arr[0,6]); worst=0; w400=0
    for t in range(426):
        u=torch.tensor([[arr[t+1,10],arr[t+1,11],arr[t+1,12],arr[t+1,13],0.0]])
        x=torch.hstack((s,u)); s=(x+torch.tensor(0.02)*dyn.nlplant(x))[:,:12]
        ref=arr[t+1,:9]; e=float(np.max(np.abs(s[0,:9].numpy()-ref)/np.maximum(np.abs(ref),floors))); worst=max(worst,e)
        if t<400: w400=worst
    return w400,worst
def mk(kind):
    def fwd(self,x):
        h=x.to(torch.float32)
        for layer in self.layers:
            if isinstance(layer,nn.Linear):
                W,b=layer.weight,layer.bias
                if kind=='layer_exact': h=(h.double()@W.double().T+b.double()).float()
                elif kind=='dot_exact_then_bias': h=(h.double()@W.double().T).float()+b
                elif kind=='chain_bias_first':
                    acc=b.unsqueeze(0).expand(h.shape[0],-1).clone()
                    for k in range(W.shape[1]): acc=(W[:,k].double().unsqueeze(0)*h[:,k:k+1].double()+acc.double()).float()
                    h=acc
                elif kind=='chain_zero_then_bias':
                    acc=torch.zeros(h.shape[0],W.shape[0])
                    for k in range(W.shape[1]): acc=(W[:,k].double().unsqueeze(0)*h[:,k:k+1].double()+acc.double()).float()
                    h=acc+b
                elif kind=='mul_add_no_fma_bias_last':
                    acc=torch.zeros(h.shape[0],W.shape[0])
                    for k in range(W.shape[1]): acc=acc+W[:,k].unsqueeze(0)*h[:,k:k+1]
                    h=acc+b
                elif kind=='mul_add_no_fma_bias_first':
                    acc=b.unsqueeze(0).expand(h.shape[0],-1).clone()
                    for k in range(W.shape[1]): acc=acc+W[:,k].unsqueeze(0)*h[:,k:k+1]
                    h=acc
            else: h=torch.relu(h)
        return h.reshape(-1)
    return fwd
print('aten', replay())
for kind in ('layer_exact','dot_exact_then_bias','chain_bias_first','chain_zero_then_bias','mul_add_no_fma_bias_last','mul_add_no_fma_bias_first'):
    mlp_cls.forward=mk(kind)
    print(kind, replay(), flush=True)
mlp_cls.forward=o_fwd
