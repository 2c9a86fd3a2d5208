import sys; sys.path.insert(0,'/root/repo')
from pathtracer_amd import binding as B, scene as S
for name, sc, d in (("sample", S.sample_scene(1920,1080,camera_width=5.0*1920/1080), 4), ("room", S.diffuse_room_scene(1920,1080), 16)):
    with B.PtGpu(max_depth=d, concurrent_passes=1) as g:
        g.upload_scene(sc); g.render(16); st=g.stats()
        P=1920*1080*16
        print(name, "queue_gib", round(st["queue_bytes"]/2**30,2), "growth", st["queue_growth"], [round(x/P,3) for x in st["level_records"][:d+1]])
