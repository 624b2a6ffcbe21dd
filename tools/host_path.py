import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
rta = importlib.import_module("ray-tracer_amd")
for name in ("glass_spheres", "fresnel"):
    s = rta.Sample(name, 1920, 1080)
    scene = rta.Scene(s.world)
    scene.render(s.camera)
    t = time.perf_counter()
    for _ in range(10):
        canvas, st = scene.render(s.camera)
    dt = (time.perf_counter() - t) / 10
    print(name, "rtx_render (host canvas) %.3f ms wall per frame, device %.3f ms, %.0f Mrays/s PCIe-inclusive" % (dt * 1e3, st.ms_total, st.rays_total / dt / 1e6))
    t = time.perf_counter()
    for _ in range(3):
        canvas, c = rta.render(s.camera, s.world)
    dt = (time.perf_counter() - t) / 3
    print(name, "rt_render (by-value drop-in: flatten + upload + render + copy) %.3f ms wall per frame" % (dt * 1e3))
    scene.close()
