import math, torch, time
dev=torch.device('cuda:0')
def smooth(n):
    while True:
        m=n
        for p in (2,3,5,7):
            while m%p==0: m//=p
        if m==1: return n
        n+=1
def t(h,w,b=24,it=5):
    x=torch.randn(b,h,w,dtype=torch.complex64,device=dev)
    for _ in range(2): torch.fft.ifft2(x)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(it): torch.fft.ifft2(x)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/it*1e3
H,W=1080,1920
tot=[0,0,0]
for k in range(15):
    h=math.ceil(H/2**(k/2)-1e-9); w=math.ceil(W/2**(k/2)-1e-9)
    hs,ws=smooth(h),smooth(w)
    he,we=(h+7)//8*8,(w+7)//8*8
    a,b,c=t(h,w),t(hs,ws),t(he,we)
    tot[0]+=a;tot[1]+=b;tot[2]+=c
    print(k,(h,w),'%.3f'%a,(hs,ws),'%.3f'%b,(he,we),'%.3f'%c)
print('total ms per 24-band analysis-equivalent', tot)
