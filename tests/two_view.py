"""Synthetic two-view correspondences (pinhole camera, known motion) with noise and gross outliers, for the
fundamental-matrix tests."""
import numpy as np
def two_view(n, outlier_frac, noise, seed, w=752, h=480, f=460.0):
    rng = np.random.default_rng(seed)
    K = np.array([[f,0,w/2],[0,f,h/2],[0,0,1.0]])
    X = np.column_stack([rng.uniform(-4,4,n*3), rng.uniform(-3,3,n*3), rng.uniform(4,12,n*3)])
    ang = 0.05; R = np.array([[np.cos(ang),0,np.sin(ang)],[0,1,0],[-np.sin(ang),0,np.cos(ang)]])
    t = np.array([0.3, 0.05, 0.1])
    x1 = (K @ X.T).T; x1 = x1[:,:2]/x1[:,2:]
    X2 = (R @ X.T).T + t; x2 = (K @ X2.T).T; x2 = x2[:,:2]/x2[:,2:]
    ok = (x1[:,0]>0)&(x1[:,0]<w)&(x1[:,1]>0)&(x1[:,1]<h)&(x2[:,0]>0)&(x2[:,0]<w)&(x2[:,1]>0)&(x2[:,1]<h)
    x1, x2 = x1[ok][:n], x2[ok][:n]
    n = len(x1)
    x1 = x1 + rng.normal(0, noise, x1.shape); x2 = x2 + rng.normal(0, noise, x2.shape)
    out = rng.random(n) < outlier_frac
    x2[out] += rng.uniform(-40, 40, (out.sum(), 2)) + np.sign(rng.normal(size=(out.sum(),2)))*8
    tx = np.array([[0,-t[2],t[1]],[t[2],0,-t[0]],[-t[1],t[0],0]])
    Ftrue = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    return x1.astype(np.float32), x2.astype(np.float32), ~out, Ftrue
