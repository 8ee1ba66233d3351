import os, subprocess, sys
ROOT='/root/repo' if os.path.exists('/root/repo/__graft_entry__.py') else os.getcwd()
if len(sys.argv)>1 and sys.argv[1]=='--child':
    import numpy as np
    sys.path.insert(0,ROOT)
    import __graft_entry__ as g
    pkg=g.load_package(); P=pkg.problems
    for N in (60,80,100,120):
        prob=P.quadrotor(N); batch=8192
        s=pkg.TinyMPC()
        s.setup(prob.A,prob.B,prob.Q,prob.R,prob.N,batch=batch,rho=prob.rho,max_iter=100,abs_pri_tol=0.0,abs_dua_tol=0.0)
        s.set_bound_constraints(prob.x_min,prob.x_max,prob.u_min,prob.u_max)
        s.set_x0_batch(np.asfortranarray(prob.x0[:,None]*np.linspace(0.5,1.0,batch)[None,:])); s.prepare()
        ms=[]
        for _ in range(5):
            s.reset_workspace(); ms.append(s.solve_timed())
        t=float(np.median(ms[2:]))
        print(f"quadrotor N={N:3d} x{batch} layout {s.launch_info()['layout']} {t:7.3f} ms {batch*100/t/1e3:7.1f} M iters/s  {s.jit_info()[:70]}",flush=True)
        s.reset()
    sys.exit(0)
for lay in (None,"E"):
    env=dict(os.environ); env.pop("TINYMPC_LAYOUT",None)
    if lay: env["TINYMPC_LAYOUT"]=lay
    print("---- TINYMPC_LAYOUT=%s"%lay,flush=True)
    subprocess.run([sys.executable,os.path.abspath(__file__),"--child"],env=env)
