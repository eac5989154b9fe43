#!/usr/bin/env python
"""Executable model of csrc/seanet_uptail.hip's schedule at lane level (numpy, float64, no operand splitting): the same tile walk,
wave / lane -> element mappings of the 32x32x16 and 16x16x32 MFMA operands and results, LDS row formulas (x tiles by parity, the
h / h' / y tiles with the y rows grouped by sample phase), carried rows, warm-up tile and phase order (the transposed convolution
running ahead: substeps 0-9 of tile k+2 in I3, 10-11 / 12-15 of tile k+1 in I1 / I2) - against the plain layer formulas.  It was
written before the kernel first ran (the index arithmetic was debugged here, not on the GPU) and is kept as the description of
the layout; tests/test_uptail_model.py runs it on the CPU.  usage: uptail_layout_model.py [T tiles] ..."""
import sys

import numpy as np

rng = np.random.default_rng(0)
FC=128; FTI=32; FTS=128; FXR=33; FHR=130
def elu(v): return np.where(v>0, v, np.exp(np.minimum(v,0))-1)

def mfma32(A,B,acc):  # A,B: [64][8], acc [64][16]
    Am=np.zeros((32,16)); Bm=np.zeros((16,32))
    for lane in range(64):
        i=lane&31; g=lane>>5
        Am[i,8*g:8*g+8]=A[lane]; Bm[8*g:8*g+8,i]=B[lane]
    D=Am@Bm
    out=acc.copy()
    for lane in range(64):
        j=lane&31; g=lane>>5
        for r in range(16):
            i=8*(r>>2)+4*g+(r&3)
            out[lane,r]+=D[i,j]
    return out
def mfma16(A,B,acc):  # A,B [64][8], acc [64][4]
    Am=np.zeros((16,32)); Bm=np.zeros((32,16))
    for lane in range(64):
        i=lane&15; q=lane>>4
        Am[i,8*q:8*q+8]=A[lane]; Bm[8*q:8*q+8,i]=B[lane]
    D=Am@Bm
    out=acc.copy()
    for lane in range(64):
        j=lane&15; q=lane>>4
        for r in range(4):
            out[lane,r]+=D[4*q+r,j]
    return out

def kernel_wg(x, wu, bu, w1, b1, w2, b2, wf, bf, wav, T, tiles, bx):
    ntile=(T+FTI-1)//FTI; tile0=bx*tiles
    if tile0>=ntile: return
    kfirst=tile0-1 if tile0>0 else 0
    klast=min(tile0+tiles,ntile)-1
    S=4*T
    xs=np.full((2,FXR,FC),np.nan); hs_h=np.full((FHR,64),np.nan); hp=np.full((FHR,64),np.nan)
    ys=np.full((FTS,32),np.nan)
    chs=np.zeros((2,64)); cps=np.zeros((2,64))
    def stage_piece(buf,t0,q):
        for tid in range(512):
            idx=tid+q*512; rr=idx>>5; cc=idx&31
            p=t0+rr; pc=p if (rr<FXR and p<=T) else 0
            if rr<FXR: xs[buf,rr,cc*4:cc*4+4]=x[pc,cc*4:cc*4+4]
    for q in range(3): stage_piece(kfirst&1,kfirst*FTI,q)
    for q in range(3): stage_piece((kfirst+1)&1,(kfirst+1)*FTI,q)
    accP=np.zeros((8,64,16)); hraw=np.zeros((8,64,16))
    def up_part(xbuf,lo,hi):
        for wave in range(8):
            for s in range(lo,hi):
                A=np.zeros((64,8)); B=np.zeros((64,8))
                for lane in range(64):
                    frow=lane&31; fg=lane>>5
                    row=frow+(s>>3); ch0=(s&7)*16+fg*8
                    A[lane]=xs[xbuf,row,ch0:ch0+8]
                    B[lane]=wu[wave*32+frow, s*16+fg*8: s*16+fg*8+8]
                assert not np.isnan(A).any()
                accP[wave]=mfma32(A,B,accP[wave])
    def h_raw():
        for wave in range(8):
            for lane in range(64):
                frow=lane&31
                hraw[wave,lane,:]=accP[wave,lane,:]+bu[wave*32+frow]
            accP[wave]=0
    def h_elu():
        hs_h[:]=np.nan
        for wave in range(8):
            ph=wave>>1; cbase=32*(wave&1)
            for lane in range(64):
                frow=lane&31; fg=lane>>5
                for q in range(16):
                    t=8*(q>>2)+4*fg+(q&3)
                    hs_h[2+4*t+ph, cbase+frow]=elu(hraw[wave,lane,q])
    def conv1():
        for wave in range(8):
            acc=[np.zeros((64,4)),np.zeros((64,4))]
            for s in range(6):
                A=np.zeros((64,8))
                for lane in range(64):
                    col=lane&15; kq=lane>>4
                    row=16*wave+col+(s>>1); c0=(s&1)*32+kq*8
                    A[lane]=hs_h[row,c0:c0+8]
                assert not np.isnan(A).any()
                for nt in range(2):
                    B=np.zeros((64,8))
                    for lane in range(64):
                        col=lane&15; kq=lane>>4
                        B[lane]=w1[16*nt+col, s*32+kq*8:s*32+kq*8+8]
                    acc[nt]=mfma16(A,B,acc[nt])
            for nt in range(2):
                for lane in range(64):
                    col=lane&15; kq=lane>>4
                    for i in range(4):
                        ys[i*32+4*wave+kq, 16*nt+col]=elu(acc[nt][lane,i]+b1[16*nt+col])
    def conv2():
        hp[2:]=np.nan
        for wave in range(8):
            ph=wave>>1; cbase=32*(wave&1)
            acc2=np.zeros((64,16))
            for s in range(2):
                A=np.zeros((64,8)); B=np.zeros((64,8))
                for lane in range(64):
                    frow=lane&31; fg=lane>>5
                    A[lane]=ys[ph*32+frow, s*16+fg*8:s*16+fg*8+8]
                    B[lane]=w2[(wave&1)*32+frow, s*16+fg*8:s*16+fg*8+8]
                acc2=mfma32(A,B,acc2)
            for lane in range(64):
                frow=lane&31; fg=lane>>5
                for q in range(16):
                    t=8*(q>>2)+4*fg+(q&3)
                    hp[2+4*t+ph, cbase+frow]=elu(hraw[wave,lane,q]+acc2[lane,q]+b2[cbase+frow])
    def conv3(s0,store):
        for wave in range(8):
            for i in range(16):
                tot=0.0
                for lane in range(64):
                    tot+=wf[0,lane]*hp[16*wave+i,lane]+wf[1,lane]*hp[16*wave+i+1,lane]+wf[2,lane]*hp[16*wave+i+2,lane]
                sidx=s0+16*wave+i
                if store and sidx<S: wav[sidx]=tot+bf
    # prologue
    up_part(kfirst&1,0,16); h_raw(); h_elu(); hs_h[0:2]=chs
    up_part((kfirst+1)&1,0,10)
    for k in range(kfirst,klast+1):
        nxt=k<klast; store=k>=tile0
        x1=(k+1)&1; x2=k&1; t2=(k+2)*FTI
        stage_piece(x2,t2,0)
        up_part(x1,10,12); conv1(); chs[:]=hs_h[FTS:FTS+2]
        stage_piece(x2,t2,1); stage_piece(x2,t2,2)
        up_part(x1,12,16); conv2(); hp[0:2]=cps
        if nxt:
            h_raw(); conv3(k*FTS,store); h_elu(); up_part(x2,0,10); hs_h[0:2]=chs
        else:
            conv3(k*FTS,store)
        cps[:]=hp[FTS:FTS+2]
        if not nxt: break

def reference(x, wu, bu, w1, b1, w2, b2, wf, bf, T):
    # x [1+T][128] with zero row first
    S=4*T
    h=np.zeros((S,64))
    for t in range(T):
        A=np.concatenate([x[t],x[t+1]])
        row=wu@A+bu
        for r in range(4): h[4*t+r]=row[r*64:(r+1)*64]
    eh=np.concatenate([np.zeros((2,64)),elu(h)])
    y=np.zeros((S,32))
    for s in range(S):
        y[s]=w1@np.concatenate([eh[s],eh[s+1],eh[s+2]])+b1
    hp=h+elu(y)@w2.T+b2
    ehp=np.concatenate([np.zeros((2,64)),elu(hp)])
    out=np.zeros(S)
    for s in range(S):
        out[s]=(wf[0]*ehp[s]).sum()+(wf[1]*ehp[s+1]).sum()+(wf[2]*ehp[s+2]).sum()+bf
    return out



def run(T, tiles, seed=0):
    """max |model - layers| for one utterance of T input rows, `tiles` tiles per workgroup."""
    g = np.random.default_rng(seed)
    x = np.zeros((1 + T + 4, 128)); x[1:1 + T] = g.standard_normal((T, 128))
    wu = g.standard_normal((256, 256)) * 0.06; bu = np.tile(g.standard_normal(64) * 0.1, 4)
    w1 = g.standard_normal((32, 192)) * 0.07; b1 = g.standard_normal(32) * 0.1
    w2 = g.standard_normal((64, 32)) * 0.17; b2 = g.standard_normal(64) * 0.1
    wf = g.standard_normal((3, 64)) * 0.07; bf = 0.03
    wav = np.full(4 * T, np.nan)
    ntile = (T + 31) // 32
    for bx in range((ntile + tiles - 1) // tiles):
        kernel_wg(x, wu, bu, w1, b1, w2, b2, wf, bf, wav, T, tiles, bx)
    ref = reference(x, wu, bu, w1, b1, w2, b2, wf, bf, T)
    return float(np.nanmax(np.abs(wav - ref))), int(np.isnan(wav).sum())


if __name__ == "__main__":
    cases = [(70, 1), (70, 2), (33, 5), (100, 2)]
    if len(sys.argv) > 2:
        cases = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for T, tiles in cases:
        err, nans = run(T, tiles)
        print(f"T = {T:4d} input rows, {tiles} tile(s) per workgroup: max |model - layers| = {err:.2e}, samples never written: {nans}")
