import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from aprilslam_amd import _lib, synth
det = _lib.Detector(id_limit=0, decimate=2.0)
dev = torch.device("cuda", 0)
B = 256
d_frames, _, _ = bench.render_stream_device(det, B, dev)
K = synth.camera_matrix(bench.W, bench.H)
st = torch.cuda.current_stream(dev).cuda_stream
det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
det.collect(max_per_frame=bench.MAXDET)
cl = det.debug_clusters()
n = cl[:, 1].astype(np.int64)
print("clusters", len(n), "per frame", len(n) / B)
edges = [24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 1 << 30]
h, _ = np.histogram(n, bins=edges)
for a, b, c in zip(edges[:-1], edges[1:], h): print("%5d..%-6d %7d  %.1f%%  points %.1f%%" % (a, b - 1, c, 100 * c / len(n), 100 * n[(n >= a) & (n < b)].sum() / n.sum()))
