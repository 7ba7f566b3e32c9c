import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, torch, helpers
case='collecthealth_s13'
s0,tr,meta,obs=helpers.load_case(case)
frames=sorted(obs); scenes=[helpers.frame_scene(s0,obs[f]) for f in frames]
for rep in range(4):
    eng=helpers.make_engine_for_scene(s0,len(scenes),agent_radius=float(meta.get("agent_radius",0.4)))
    eng.set_state(helpers.scene_state_arrays(scenes))
    for k in range(3):
        rgb=torch.zeros((len(scenes),60,80,3),dtype=torch.uint8,device="cuda")
        eng.render_top(rgb,None,True); eng.check()
        r=rgb.cpu().numpy()
        for i,f in enumerate(frames):
            d=np.any(r[i]!=obs[f]["top_rgb"],axis=2)
            if d.any():
                ys,xs=np.nonzero(d)
                print("rep",rep,"call",k,"frame",i,f,"npx",d.sum(),"x",xs.min(),xs.max(),"y",ys.min(),ys.max(), "tiles", sorted(set((int(y)//4)*5+int(x)//16 for y,x in zip(ys,xs))))
    eng.close()
print("n frames", len(frames))
