import itertools
G = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
     list(range(4,12))+list(range(16,20))+list(range(28,32)),
     list(range(32,36))+list(range(44,48))+list(range(52,60)),
     list(range(36,44))+list(range(48,52))+list(range(60,64))]
def cycles(addrs):
    tot=0
    for grp in G:
        banks={}
        for l in grp:
            a=addrs[l]
            for dw in range(4):
                banks.setdefault(((a//4)+dw)%64,set()).add(a)
        tot+=max(len(v) for v in banks.values())
    return tot
def conv_sim(CC, TH, TW, POOL, PWL, swz=None, WM=4):
    BM=TH*TW; MT_TOT=(BM+15)//16
    SPC = 3 if CC==16 else 5
    res=[]
    for mt in range(MT_TOT):
        for t in range(SPC):
            addrs=[]
            for l in range(64):
                g=l>>4; li=l&15
                row=min(mt*16+li,BM-1)
                if POOL:
                    w=row>>2; r=row&3
                    oy=2*(w//(TW//2))+(r>>1); ox=2*(w%(TW//2))+(r&1)
                else:
                    oy=row//TW; ox=row%TW
                if CC==16:
                    tap=min(4*t+g,8); half=0
                else:
                    tap=min(2*t+(g>>1),8); half=g&1
                y=oy+tap//3; x=ox+tap%3
                a=(y*PWL+x)*CC+half*16
                if swz: a=swz(a,y,x,half,CC)
                addrs.append(a)
            res.append(cycles(addrs))
    return sum(res)/len(res)
print("conv2 CC=16 16x52 pool")
for PWL in range(54,72): print(PWL, round(conv_sim(16,16,52,True,PWL),2), end=" | ")
print()
print("conv3_1 CC=32 13x26")
for PWL in range(28,44): print(PWL, round(conv_sim(32,13,26,False,PWL),2), end=" | ")
print()
# swizzles for CC=32: xor half with row parity / with x bits
def sw1(a,y,x,h,CC): return a ^ ((y&1)<<4)
def sw2(a,y,x,h,CC): return a ^ (((x>>1)&1)<<4)
def sw3(a,y,x,h,CC): return a ^ (((x>>2)&1)<<4)
for name,sw in (("ypar",sw1),("x>>1",sw2),("x>>2",sw3)):
    print("conv3_1 swz",name)
    for PWL in range(28,44): print(PWL, round(conv_sim(32,13,26,False,PWL,sw),2), end=" | ")
    print()
