import ctypes, torch, sys
sys.path.insert(0,'/root/repo')
from moleculardiffusion_mivit_amd import _native as N
p=lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g=torch.Generator(device='cuda').manual_seed(0)
for fam in ('rowstream','wavestream'):
  for (M,Nn,K,var) in [(264,128,256,'dact'),(264,256,128,'dres'),(264,128,128,'plain'),(4133,128,256,'dact')]:
    dy=torch.randn(M,Nn,device='cuda',generator=g)*30; W=torch.randn(Nn,K,device='cuda',generator=g)*0.1
    saved=torch.randn(M,K,device='cuda',generator=g); dres=torch.randn(M,K,device='cuda',generator=g)
    dyb,Wb,sb,rb=dy.bfloat16(),W.bfloat16(),saved.bfloat16(),dres.bfloat16()
    dx=torch.empty(M,K,dtype=torch.bfloat16,device='cuda')
    entry=getattr(N.lib,f'mivit_{fam}_dgrad')
    rc=entry(p(dyb),Nn,p(Wb),M,Nn,K,1 if var=='dact' else 0,p(sb) if var=='dact' else None,K,p(rb) if var=='dres' else None,K,p(dx),K,st)
    ref=dyb.double()@Wb.double()
    if var=='dact': ref=ref*(sb.double()>0)
    if var=='dres': ref=ref+rb.double()
    err=(dx.double()-ref).abs().max()/ref.abs().max()
    print(fam,M,Nn,K,var,'rc',rc,'err %.2e'%float(err))
