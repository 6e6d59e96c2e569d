#!/usr/bin/env python3
"""Diagnostic: the same batch many times through one detector; any repetition whose results differ from the first one is
reported field by field (a difference between repetitions is a race or an uninitialised read).

    ASL_LIB=build/libaprilslam_x.so python tools/race_hunt.py [--batch 1024] [--reps 40]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
from aprilslam_amd import _lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--clutter", type=int, default=0, help="instead of the bench scene: a batch of this many cluttered 720p frames (tools/fuzz_clutter.py: noise, stripes, "
                    "checkerboards over tag scenes -- the dense launches, the large size classes, buffers that grow)")
    ap.add_argument("--refit", type=int, default=0, help="after one batch, re-run only the quad fit this many times (asl_debug_fetch item 7)")
    ap.add_argument("--quads", action="store_true", help="compare the quads of every repetition; on a difference also the clusters")
    ap.add_argument("--deep", action="store_true", help="also compare labels, sizes, clusters (sorted points) and quads of every repetition")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.batch
    det = _lib.Detector(id_limit=0, decimate=2.0)
    if args.clutter:
        import fuzz_clutter
        rng = np.random.default_rng(4711)
        B = args.clutter
        frames = []
        for _ in range(B):
            tags = synth.random_scene(bench.W, bench.H, int(rng.integers(1, 6)), rng)
            f, _ = synth.render_frame(bench.W, bench.H, tags, 18.0, cam_position=tuple(rng.uniform(-2, 2, 3)), cam_rotation_deg=tuple(rng.uniform(-3, 3, 3)))
            frames.append(fuzz_clutter.clutter(f, rng))
        d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    else:
        d_frames, _, _ = bench.render_stream_device(det, B, dev)
    K = synth.camera_matrix(bench.W, bench.H)
    st = torch.cuda.current_stream(dev).cuda_stream
    if args.refit:
        det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
        det.collect(max_per_frame=bench.MAXDET)
        r = det.debug_refit(args.refit)
        print("refit: %d repetitions, %d quads differ from the first one, by size class %s" % (r[0], r[1], r[2:].tolist()))
        return
    first = None
    nbad = 0
    for it in range(args.reps):
        det.submit_device(d_frames.data_ptr(), B, 3, bench.W, bench.H, stream=st, K=K, dist=np.zeros(4), tag_size=bench.TAG_INNER)
        dets, poses, npf = det.collect(max_per_frame=max(bench.MAXDET, 256))
        cur = (np.array(dets, copy=True), np.array(poses, copy=True), np.array(npf, copy=True), det.debug_counters().tolist())
        if args.quads:
            q = det.debug_quads(cap=1 << 20).copy()
            if first is None:
                first_q, first_cl = q, det.debug_clusters().copy()
            elif q.tobytes() != first_q.tobytes():
                cl = det.debug_clusters().copy()
                same_cl = cl.shape == first_cl.shape and cl.tobytes() == first_cl.tobytes()
                ka = {(int(f), int(c)): i for i, (f, c) in enumerate(zip(first_q["frame"], first_q["cluster"]))}
                kb = {(int(f), int(c)): i for i, (f, c) in enumerate(zip(q["frame"], q["cluster"]))}
                lost, gained = sorted(set(ka) - set(kb)), sorted(set(kb) - set(ka))
                moved = [k for k in ka if k in kb and first_q["p"][ka[k]].tobytes() != q["p"][kb[k]].tobytes()]
                def npts(k):
                    key = (k[0] << 48) | ((k[1] >> 32) << 24) | (k[1] & 0xFFFFFF)
                    hit = np.nonzero(first_cl[:, 0] == np.uint64(key))[0]
                    return int(first_cl[hit[0], 1]) if len(hit) else -1
                print("rep %d: quads differ, clusters %s; lost %s gained %s moved %s" % (
                    it, "identical" if same_cl else "DIFFER", [(k, npts(k)) for k in lost[:4]], [(k, npts(k)) for k in gained[:4]], [(k, npts(k)) for k in moved[:4]]))
        if args.deep:
            stages = {"labels": det.debug_image(2), "sizes": det.debug_image(3), "clusters": det.debug_clusters().copy(), "quads": det.debug_quads(cap=1 << 20).copy()}
            if first is None:
                first_stages = stages
            else:
                for name, arr in stages.items():
                    ref = first_stages[name]
                    if arr.shape != ref.shape or arr.tobytes() != ref.tobytes():
                        print("rep %d: stage '%s' differs from rep 0 (%s vs %s)" % (it, name, arr.shape, ref.shape))
                        if name in ("labels", "sizes") and arr.shape == ref.shape:
                            bad = np.argwhere(arr != ref)
                            print("   %d pixels, first (frame, y, x) %s: %s -> %s" % (len(bad), bad[0].tolist(), ref[tuple(bad[0])], arr[tuple(bad[0])]))
                        if name == "clusters" and arr.shape == ref.shape:
                            bad = np.nonzero((arr != ref).any(axis=1))[0]
                            print("   %d clusters, first: %s -> %s" % (len(bad), [hex(int(v)) for v in ref[bad[0]]], [hex(int(v)) for v in arr[bad[0]]]))
                        if name == "quads":
                            ka = set(zip(ref["frame"].tolist(), ref["cluster"].tolist())); kb = set(zip(arr["frame"].tolist(), arr["cluster"].tolist()))
                            print("   only in rep 0: %s   only now: %s" % (sorted(ka - kb)[:4], sorted(kb - ka)[:4]))
                            if arr.shape == ref.shape:
                                bad = np.nonzero((arr["p"] != ref["p"]).reshape(len(arr), -1).any(axis=1))[0]
                                if len(bad): print("   corners differ in %d quads, first: frame %d %s -> %s" % (len(bad), ref["frame"][bad[0]], ref["p"][bad[0]].tolist(), arr["p"][bad[0]].tolist()))
                        break
        if first is None:
            first = cur
            print("rep 0: %d detections, counters %s" % (len(dets), cur[3]))
            continue
        same = cur[0].tobytes() == first[0].tobytes() and cur[1].tobytes() == first[1].tobytes() and cur[2].tobytes() == first[2].tobytes()
        if same:
            continue
        nbad += 1
        print("rep %d differs: %d detections (first %d), counters %s" % (it, len(cur[0]), len(first[0]), cur[3]))
        fr = np.nonzero(cur[2] != first[2])[0]
        print("  frames with another detection count:", fr[:10].tolist(), [(int(first[2][f]), int(cur[2][f])) for f in fr[:10]])
        if len(cur[0]) == len(first[0]):
            for name in first[0].dtype.names or ():
                a, b = first[0][name], cur[0][name]
                bad = np.nonzero((a != b).reshape(len(a), -1).any(axis=1))[0]
                if len(bad):
                    print("  field %s differs in %d records, first %d: %s -> %s" % (name, len(bad), bad[0], a[bad[0]].tolist(), b[bad[0]].tolist()))
            if cur[1].tobytes() != first[1].tobytes():
                a, b = np.asarray(first[1]), np.asarray(cur[1])
                if a.dtype.names:
                    for name in a.dtype.names:
                        bad = np.nonzero((a[name] != b[name]).reshape(len(a), -1).any(axis=1))[0]
                        if len(bad):
                            print("  pose field %s differs in %d records, first %d" % (name, len(bad), bad[0]))
        else:
            off = np.concatenate([[0], np.cumsum(first[2])]); off2 = np.concatenate([[0], np.cumsum(cur[2])])
            for f in fr[:3]:
                print("  frame %d first:" % f, first[0][off[f]:off[f + 1]][["id", "hamming"]].tolist() if first[0].dtype.names else "")
                print("  frame %d now:  " % f, cur[0][off2[f]:off2[f + 1]][["id", "hamming"]].tolist() if cur[0].dtype.names else "")
    print("%d of %d repetitions differ from the first" % (nbad, args.reps - 1))


if __name__ == "__main__":
    main()
