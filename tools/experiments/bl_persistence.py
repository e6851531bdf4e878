"""Diagnostic: is slow boundary-layer convergence persistent in time for a point?  If it is,
sorting the points of a workgroup by their recent trip count puts the slow ones into the same
wavefront and the other wavefronts stop after 5 iterations."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_helpers as oh
from roadsurf_amd import abi

n, L = 1024, 5761
lib = oh.load("port")
C.c_int.in_dll(lib, "oracle_bl_hist_on").value = 1
f = oh.synth_forcing(n, L, seed=20240110)
s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
trace = np.zeros((n, 2 * (L + 8)), np.uint8)
out = {k: np.empty((1, L)) for k in oh.F64_OUT}
lib.runsimulation.argtypes = [C.POINTER(abi.OutputPointers), C.POINTER(abi.InputPointers),
                              C.POINTER(abi.InputSettings), C.POINTER(abi.InputParameters),
                              C.POINTER(abi.LocalParameters)]
cnt = np.zeros(n, np.int64)
for q in range(n):
    ip, op, keep = oh.point_pointers(f, q, None) if False else (None, None, None)
    ip, op, keep = oh.point_pointers(f, q)
    C.c_void_p.in_dll(lib, "oracle_bl_trace").value = trace[q].ctypes.data
    C.c_long.in_dll(lib, "oracle_bl_trace_pos").value = 0
    C.c_long.in_dll(lib, "oracle_bl_trace_cap").value = 2 * (L + 8)
    lib.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    cnt[q] = C.c_long.in_dll(lib, "oracle_bl_trace_pos").value
print("calls per point", cnt.min(), cnt.max())
cnt //= 2                   # two bytes per call: passes, passes through the unstable branch
off = int(cnt.min()) - L   # calls before the time loop (initialisation)
t = trace[:, 2 * off:2 * (off + L):2].astype(np.int32)          # [point][step]
u = trace[:, 2 * off + 1:2 * (off + L) + 1:2].astype(np.int32)  # unstable passes of the step
np.save(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "bl_unstable.npy"), u)
print("share of point-steps with no unstable pass: %.3f" % (u == 0).mean())
extra = t - 5
print("mean trip %.3f" % t.mean())

def wave_cost(tt):      # sum over steps of the max over each group of 64 points
    g = tt.reshape(tt.shape[0] // 64, 64, tt.shape[1])
    return g.max(axis=1).mean()

print("iterations per wave-step, points in given order      : %.3f" % wave_cost(t))
CH = 240
acc = []
for bsz in (256, 1024):
    tot = 0.0; nchunks = 0
    for c0 in range(0, L - CH, CH):
        prev = extra[:, max(0, c0 - CH):c0].sum(axis=1) if c0 else np.zeros(n)
        cur = t[:, c0:c0 + CH]
        cost = 0.0
        for b0 in range(0, n, bsz):
            order = np.argsort(prev[b0:b0 + bsz], kind="stable") + b0
            cost += wave_cost(cur[order]) * (bsz // 64)
        tot += cost / (n // 64); nchunks += 1
    print(f"sorted inside groups of {bsz} by the previous {CH}-step chunk: {tot / nchunks:.3f}")
# oracle-knowledge bound: sort by the CURRENT chunk's own total
tot = 0.0; nchunks = 0
for c0 in range(0, L - CH, CH):
    cur = t[:, c0:c0 + CH]
    order = np.argsort(extra[:, c0:c0 + CH].sum(axis=1), kind="stable")
    tot += wave_cost(cur[order]); nchunks += 1
print("sorted globally by the chunk's own total (bound)       : %.3f" % (tot / nchunks))

os.makedirs(os.path.join(os.path.dirname(__file__), "..", "gpurun_out"), exist_ok=True)
np.save(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "bl_trace.npy"), t)
for CH in (60, 120, 240, 480):
    for hist in (1, 2):
        tot = 0.0; nchunks = 0
        for c0 in range(CH, L - CH, CH):
            prev = extra[:, max(0, c0 - hist * CH):c0].sum(axis=1)
            cur = t[:, c0:c0 + CH]
            cost = 0.0
            for b0 in range(0, n, 256):
                order = np.argsort(prev[b0:b0 + 256], kind="stable") + b0
                cost += wave_cost(cur[order]) * 4
            tot += cost / (n // 64); nchunks += 1
        print(f"chunk {CH:3d}, key = extra iterations over the previous {hist} chunk(s), groups of 256: {tot / nchunks:.3f}")
# key = trip count of the LAST step of the previous chunk
for CH in (60, 240):
    tot = 0.0; nchunks = 0
    for c0 in range(CH, L - CH, CH):
        prev = t[:, c0 - 1]
        cur = t[:, c0:c0 + CH]
        cost = 0.0
        for b0 in range(0, n, 256):
            order = np.argsort(prev[b0:b0 + 256], kind="stable") + b0
            cost += wave_cost(cur[order]) * 4
        tot += cost / (n // 64); nchunks += 1
    print(f"chunk {CH:3d}, key = trip count of the previous step, groups of 256: {tot / nchunks:.3f}")
