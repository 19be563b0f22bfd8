import sys, time
sys.path.insert(0, 'pbrt-v3-rs_amd')
import numpy as np, pbrt_hip, torch
h = pbrt_hip.Host()
s = pbrt_hip.Scene(); pbrt_hip.capture_spec(pbrt_hip.SceneSpec(n_tris=100000, xres=512, yres=512, spp=64), s, h)
dev = torch.device('cuda', 0)
for parts in (1, 2, 4, 8):
    floats = s.tile_buffer_floats(16, 0, parts)
    buf = torch.zeros(floats, dtype=torch.float32, device=dev); torch.cuda.synchronize()
    s.render_path_tiles_device(buf.data_ptr(), tile_part=0, tile_parts=parts)
    t = time.time(); n = 5
    for _ in range(n): st = s.render_path_tiles_device(buf.data_ptr(), tile_part=0, tile_parts=parts)
    dt = (time.time() - t) / n
    print('tile_parts', parts, 'one rank share: %.2f ms wall, %.2f ms device; ideal %.2f' % (dt * 1e3, st.render_seconds * 1e3, 47.1 / parts))
