import sys, os, torch, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
from simt_amd import ops, _lib as L, model_spec as ms
dev = torch.device("cuda:0")
def load(path):
    lib = C.CDLL(path)
    for name in ("simt_head_loss", "simt_head_grad", "simt_head_nblk", "simt_head_part_floats", "simt_head_hout_floats", "simt_head_keys_count"):
        fn = getattr(lib, name); sig = L.SIGNATURES[name]; fn.restype = sig[0]; fn.argtypes = sig[1]
    return lib
libs = [load(p) for p in sys.argv[1:3]]
B,H,W,h,w,Cn,K = 4,768,768,97,97,19,3
Q=Cn+K; lib=libs[1]; st=torch.cuda.current_stream().cuda_stream
g=torch.Generator().manual_seed(1)
p1=torch.zeros(B*h*w,32); p1[:,:Q]=torch.randn(B*h*w,Q,generator=g)*3; p2=torch.zeros(B*h*w,32); p2[:,:Q]=torch.randn(B*h*w,Q,generator=g)*3
fx=torch.zeros(B*h*w,32); fx[:,:Cn]=torch.softmax(torch.randn(B*h*w,Cn,generator=g)*4,1)
p1,p2,fx=p1.to(dev),p2.to(dev),fx.to(dev)
_,lab=ms.synthetic_batch(B,H,W,ms.load_class_dist(),seed=3,device=dev)
T=[torch.softmax(torch.randn(Q,Cn,generator=g),1).to(dev) for _ in range(2)]
nblk=max(l.simt_head_nblk(B,H,W) for l in libs)      # (the libraries may size the partial-sum rows differently)
part=torch.zeros(nblk,lib.simt_head_part_floats(Q,Cn),device=dev); keys=torch.zeros(lib.simt_head_keys_count(),device=dev,dtype=torch.int64)
hout=torch.zeros(lib.simt_head_hout_floats(Q,Cn),device=dev); g1=torch.zeros(2,B,H,w,24,device=dev)
d1=torch.zeros(B*h*w,64,device=dev,dtype=torch.bfloat16); d2=torch.zeros_like(d1)
hd=L.HeadDesc()
hd.pred1,hd.pred2,hd.fixp,hd.label=p1.data_ptr(),p2.data_ptr(),fx.data_ptr(),lab.data_ptr()
hd.T1,hd.T2=T[0].data_ptr(),T[1].data_ptr(); hd.part,hd.keys,hd.hout,hd.g1=part.data_ptr(),keys.data_ptr(),hout.data_ptr(),g1.data_ptr()
hd.dpred1_f32,hd.dpred2_f32,hd.dpred1_t,hd.dpred2_t=None,None,d1.data_ptr(),d2.data_ptr()
hd.B,hd.h,hd.w,hd.H,hd.W,hd.C,hd.Q=B,h,w,H,W,Cn,Q
hd.ldp,hd.ldf,hd.QP,hd.ld_f32,hd.ld_t,hd.grad_dtype=32,32,24,0,64,L.SIMT_BF16
hd.th_high,hd.th_low,hd.lambda_seg,hd.lambda_place,hd.gscale=0.8,0.2,0.1,0.1,1.0
if os.environ.get("AB_HEAD_WS") == "1":      # the trainers' form: both byte maps given (a library older than label_ws ignores the trailing field)
    conf_ws = torch.zeros(B, H, W, device=dev, dtype=torch.uint8); lab_ws = torch.zeros(B, H, W, device=dev, dtype=torch.uint8)
    hd.conf_out, hd.label_ws = conf_ws.data_ptr(), lab_ws.data_ptr()
flush = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
def t(fn,n=8):
    ts=[]
    for _ in range(n):
        flush.zero_(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    return np.median(ts)
for rnd in range(2):
    for name, l in zip("AB", libs):
        a = t(lambda: l.simt_head_loss(C.byref(hd), st)); b = t(lambda: l.simt_head_grad(C.byref(hd), st))
        print(f"{name}: head_loss {a:.0f} us   head_grad {b:.0f} us")
res=[]
for l in libs:
    l.simt_head_loss(C.byref(hd), st); l.simt_head_grad(C.byref(hd), st); torch.cuda.synchronize()
    res.append((hout.clone(), d1.float().clone(), d2.float().clone(), g1.clone()))
print("hout A", res[0][0][:14].cpu().tolist())
print("hout B", res[1][0][:14].cpu().tolist())
print("max|dhout|", (res[0][0][:16+4*Q*Cn]-res[1][0][:16+4*Q*Cn]).abs().max().item(), "max|dg1|", (res[0][3]-res[1][3]).abs().max().item(), "max|g1|", res[0][3].abs().max().item(),
      "max|dd2|", (res[0][2]-res[1][2]).abs().max().item(), "max|d2|", res[0][2].abs().max().item())
