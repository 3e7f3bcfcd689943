"""GPU-box measurement: predict / recommend timings at the sizes of the reference's published notebook runs
(examples/instacart.ipynb: predict 262,425 pairs 542 ms; recommend 9,936 users x 35k items top-10 45.6 s)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rankfm_amd import synthetic
from rankfm_amd._rankfm import _predict, _recommend
U, I, F = 10000, 35000, 50
pairs, csr = synthetic.make_interactions(U, I, 550000, seed=0)
w = synthetic.init_weights(U, I, F, seed=1, sigma=0.3)
z_u, z_i = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
args = (z_u, z_i, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])
idx = np.ascontiguousarray(pairs[:262425].astype(np.float32))
users = np.arange(9936, dtype=np.float32)
for rep in range(2):
    t0 = time.perf_counter(); s = _predict(idx, *args); t1 = time.perf_counter()
    r = _recommend(users, csr, 10, True, *args); t2 = time.perf_counter()
    print("predict %d pairs: %.1f ms   recommend %d users x %d items top-10 (filter_previous): %.1f ms  [host buffers, upload included]" % (
        len(idx), (t1 - t0) * 1e3, len(users), I, (t2 - t1) * 1e3), flush=True)

# the same calls on a resident session (model, features and item lists already in HBM: rfm_predict_device / rfm_recommend_device)
import torch
from rankfm_amd.engine import DeviceSession
sess = DeviceSession(pairs, np.ones(len(pairs), np.float32), csr.offsets, csr.items, z_u, z_i, w)
users_d = torch.from_numpy(users).to(sess.device)
idx_d = torch.from_numpy(idx).to(sess.device)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); s2 = sess.predict(idx_d, to_host=False); torch.cuda.synchronize(); t1 = time.perf_counter()
    r2 = sess.recommend(users_d, 10, True, to_host=False); torch.cuda.synchronize(); t2 = time.perf_counter()
    r3 = sess.recommend(users, 10, True); t3 = time.perf_counter()
    print("resident session: predict %.2f ms   recommend %.2f ms (device in, device out)   %.2f ms (host user ids in, host lists out)" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
assert np.array_equal(r3, r, equal_nan=True)
