import sys; sys.path.insert(0,'.')
from ti_raytrace_amd import scenes
for name, mk in (("headline 100k", lambda: scenes.synthetic(1024,1024,32,device_id=0)), ("teapot", lambda: scenes.single_model(1024,1024,32,device_id=0)), ("cornell", lambda: scenes.cornell_box(512,512,32,device_id=0))):
    ex = mk(); ex.build_scene(); ctx = ex.scene.ctx
    ctx.set_option("primary_beams_diag", 1)
    ex.integrator.render_frames(32); ctx.sync()
    st = ctx.primary_beam_stats(); R = st["rays"]
    print(name, "rays", R, "leaf steps per ray %.3f" % (st["diag_leaf_steps"]/R), "rays with > 1 step %.3f" % (st["diag_rays_more_than_one_step"]/R), "> 2 steps %.3f" % (st["diag_rays_more_than_two_steps"]/R),
          "lane slots per ray %.3f (utilisation %.3f)" % (st["diag_lane_slots"]/R, st["diag_leaf_steps"]/max(st["diag_lane_slots"],1)))
