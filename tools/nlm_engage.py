"""anlmdn engagement statistics on the synthetic speech signal (numpy, f64 cumulative sums: statistics, not bit-exact decisions):
per hop-pair step, how many of the 64 lanes (3 offsets each) own an offset whose patch distance is under the smoothing cut.
This is the measurement behind k_anlmdn_pair3 parking lanes 30..33 only."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from jivetalking_amd.synth import speech_like
sr=48000
x=speech_like(40.0,sr,seed=3).astype(np.float64)
K=288;S=96;H=2*K+1
sw=(65536.0/(4*K+2))/np.sqrt(1e-5); smooth=3.0
dthr=smooth/sw
n=len(x)
nh=(n-2*(K+S))//H; nh-=nh%2
lanes=np.zeros((64,nh//2,H),bool)
offs=list(range(-S,0))+list(range(1,S+1))
for j,d in enumerate(offs):
    dd=np.zeros(n)
    if d>0: dd[:n-d]=(x[:n-d]-x[d:])**2
    else: dd[-d:]=(x[-d:]-x[:n+d])**2
    cs=np.concatenate([[0],np.cumsum(dd)])
    i=np.arange(K+S, K+S+nh*H)
    dist=cs[i+K+1]-cs[i-K]
    e=(dist<dthr).reshape(nh//2,2,H)
    lanes[j//3]|=(e[:,0]|e[:,1])
cnt=lanes.sum(axis=0)   # engaged lanes per pair-step
print("P(cnt==0)",(cnt==0).mean())
for c in [1,2,3,4,6,8,12,16,32,64]:
    print("P(0<cnt<=%d)"%c, ((cnt>0)&(cnt<=c)).mean())
print("mean lanes when engaged",cnt[cnt>0].mean())
# per 64-step block total
tot=cnt[:, :576].reshape(cnt.shape[0],9,64).sum(axis=2)
print("block totals pct:",np.percentile(tot,[50,75,90,95,99,100]))
mx=cnt[:, :576].reshape(cnt.shape[0],9,64).max(axis=2)
print("block max pct:",np.percentile(mx,[50,75,90,95,99,100]))
print("which lanes:",np.nonzero(lanes.mean(axis=(1,2))>0.001)[0])
