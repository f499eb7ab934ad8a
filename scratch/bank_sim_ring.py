# LDS bank-conflict cycles of the ring kernel's A-fragment reads (ds_read_b128, 64-byte pixel rows, pitch PWL pixels)
# for candidate swizzles s(y, x) of the 16-byte group:  addr = (y*PWL + x)*64 + ((g ^ s(y, x)) << 4)
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
def sim(TH,TW,POOL,PWL,s):
    BM=TH*TW; MT_TOT=(BM+15)//16
    tot=0;n=0
    for mt in range(MT_TOT):
        for ky in range(3):
            for dx in range(3):
                addrs=[]
                for l in range(64):
                    g=l>>4; li=l&15
                    row=min(mt*16+li,BM-1)
                    if POOL:
                        w=row>>2; r=row&3
                        oy=2*(w//(TW//2))+(r>>1); ox=2*(w%(TW//2))+(r&1)
                    else:
                        oy=row//TW; ox=row%TW
                    y=oy+ky; x=ox+dx
                    addrs.append((y*PWL+x)*64+((g^s(y,x))<<4))
                tot+=cycles(addrs); n+=1
    return tot/n
cands={
 "cur (x>>1)&3": lambda y,x:(x>>1)&3,
 "x&3": lambda y,x:x&3,
 "(x>>2)&3": lambda y,x:(x>>2)&3,
 "((x>>1)+y)&3": lambda y,x:((x>>1)+y)&3,
 "((x>>1)+2y)&3": lambda y,x:((x>>1)+2*y)&3,
 "((x>>1)^y)&3": lambda y,x:((x>>1)^y)&3,
 "((x>>1)+3y)&3": lambda y,x:((x>>1)+3*y)&3,
 "((x+y)>>1)&3": lambda y,x:((x+y)>>1)&3,
 "((x+2y)>>1)&3": lambda y,x:((x+2*y)>>1)&3,
 "((x+3y)>>1)&3": lambda y,x:((x+3*y)>>1)&3,
 "((x>>1)+(y>>1))&3": lambda y,x:((x>>1)+(y>>1))&3,
 "(x+y)&3": lambda y,x:(x+y)&3,
 "none": lambda y,x:0,
}
for shape in ((13,26,False),(26,26,True),(13,13,False)):
    print("tile",shape,"PWL 32 (13x13: 16): cycles per ds_read_b128 (4 = conflict-free)")
    PWL = 16 if shape[1]==13 else 32
    for k,f in cands.items():
        print("   %-20s %.3f"%(k, sim(*shape,PWL,f)))

import itertools
print("brute force: s = (a*x + b*(x>>1) + c*(x>>2) + d*y + e*(y>>1)) & 3, applied as XOR or ADD to the group index")
best={}
for shape in ((13,26,False),(26,26,True),(13,13,False)):
    PWL = 16 if shape[1]==13 else 32
    res=[]
    for a,b,c,d,e in itertools.product(range(4),repeat=5):
        f=lambda y,x,a=a,b=b,c=c,d=d,e=e:(a*x+b*(x>>1)+c*(x>>2)+d*y+e*(y>>1))&3
        res.append((sim(*shape,PWL,f),(a,b,c,d,e)))
    res.sort()
    print(shape, "best 6:", [(round(v,3),k) for v,k in res[:6]])
    best[shape]=res
# common best over the three shapes (weighted by use: 13x26 x4 layers, 26x26 x2, 13x13 x1)
tot={}
for shape,w in (((13,26,False),4),((26,26,True),2),((13,13,False),1)):
    for v,k in best[shape]: tot[k]=tot.get(k,0)+w*v
print("best common:", sorted((v/7,k) for k,v in tot.items())[:6])
