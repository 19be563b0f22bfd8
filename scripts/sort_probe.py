"""Does ray ordering matter to the traversal kernel?  Bounce-like rays (origins on the geometry the camera sees, directions uniform
on the hemisphere facing back) traced in three orders: path order (a wave = consecutive samples of one pixel), a random permutation,
and sorted by direction octant + Morton code of the origin.  Measurement aid, not part of the product path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pbrt-v3-rs_amd"))
import numpy as np, pbrt_hip, torch

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
h = pbrt_hip.Host()
s = pbrt_hip.Scene(); pbrt_hip.capture_spec(pbrt_hip.SceneSpec(n_tris=n_tris, xres=512, yres=512, spp=16), s, h)
rng = np.random.default_rng(1)
cams = []
for smp in range(16):
    r, _ = s.generate_camera_rays([0, 0, 512, 512], smp); cams.append(r)
cam = np.stack(cams, axis=1).reshape(-1)   # [pixel][sample]: the renderer's path order
hits = s.intersect_batch(cam)
ok = hits["prim"] != 0xFFFFFFFF
p = cam["o"] + cam["d"] * hits["t"][:, None]
d = rng.normal(size=(len(cam), 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
flip = (d * cam["d"]).sum(1) > 0; d[flip] = -d[flip]
b = np.zeros(int(ok.sum()), pbrt_hip.RAY_DTYPE)
b["o"] = p[ok] + 1e-3 * d[ok]; b["d"] = d[ok]; b["t_max"] = np.inf
print("bounce rays:", len(b))
lo, hi = b["o"].min(0), b["o"].max(0)
q = np.clip(((b["o"] - lo) / (hi - lo) * 1023).astype(np.uint32), 0, 1023)
def spread(x):
    x = x.astype(np.uint64); x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249; return x
mort = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
octant = ((b["d"][:, 0] < 0).astype(np.uint64) | ((b["d"][:, 1] < 0).astype(np.uint64) << 1) | ((b["d"][:, 2] < 0).astype(np.uint64) << 2))
orders = {"path order": np.arange(len(b)), "random": rng.permutation(len(b)), "octant+morton": np.argsort((octant << 30) | mort, kind="stable"),
          "morton only": np.argsort(mort, kind="stable")}
dev = torch.device("cuda", 0)
out = torch.zeros(len(b) * 8, dtype=torch.float32, device=dev)
ref = None
for name, o in orders.items():
    rr = torch.from_numpy(np.ascontiguousarray(b[o]).view(np.float32).reshape(-1, 8)).to(dev); torch.cuda.synchronize()
    ms = [s.intersect_batch_device(rr.data_ptr(), out.data_ptr(), len(b)) for _ in range(6)]
    t = out.cpu().numpy().reshape(-1, 8)[:, 0].copy()
    back = np.empty_like(t); back[o] = t
    if ref is None: ref = back
    print(f"{name:16s} {min(ms[1:]):7.3f} ms  {len(b) / min(ms[1:]) / 1e3:8.1f} Mrays/s  same t: {np.array_equal(ref.view(np.uint32), back.view(np.uint32))}")
