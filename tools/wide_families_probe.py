import os, sys, numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/__graft_entry__.py') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle'))
import __graft_entry__ as g
pkg=g.load_package(); P=pkg.problems
def system(nx,nu,N,seed=0):
    rng=np.random.default_rng(seed)
    A=np.eye(nx)+0.03*rng.standard_normal((nx,nx)); B=0.1*rng.standard_normal((nx,nu))
    prob=P.Problem("wide",A,B,np.diag(rng.uniform(1,10,nx)),np.diag(rng.uniform(0.5,2,nu)),N,2.0,rng.standard_normal(nx))
    prob.u_min,prob.u_max=np.full(nu,-0.3),np.full(nu,0.3); prob.x_min,prob.x_max=np.full(nx,-2.0),np.full(nx,2.0)
    return prob
CASES=((24,8,30,4096),(48,16,20,2048),(12,4,30,8192))
if len(sys.argv)>1 and sys.argv[1]=='large':  # layout M: the families' phase next to the box path
    CASES=((96,32,20,4096),(160,32,12,2048),(300,20,8,1024))
for nx,nu,N,batch in CASES:
    prob=system(nx,nu,N)
    for fam in ((False,True,"cone") if nx>64 else (False,True)):
        s=pkg.TinyMPC()
        s.setup(prob.A,prob.B,prob.Q,prob.R,prob.N,batch=batch,rho=prob.rho,max_iter=50 if nx>64 else 100,abs_pri_tol=0.0,abs_dua_tol=0.0)
        s.set_bound_constraints(prob.x_min,prob.x_max,prob.u_min,prob.u_max)
        if fam:
            s.set_cone_constraints(Acx=[0],qcx=[3],cx=[0.7],Acu=[],qcu=[],cu=[])
            rng=np.random.default_rng(1)
            if fam is True: s.set_linear_constraints(Alin_x=rng.standard_normal((2,nx)),blin_x=np.array([1.0,1.5]),Alin_u=np.zeros((0,nu)),blin_u=np.zeros(0))
        x0s=np.asfortranarray(prob.x0[:,None]+0.1*np.random.default_rng(2).standard_normal((nx,batch)))
        s.set_x0_batch(x0s); s.prepare()
        ms=[]
        for k in range(4):
            s.reset_workspace(); ms.append(s.solve_timed())
        t=float(np.median(ms[1:]))
        print(f"nx={nx} nu={nu} N={N} batch={batch} families={fam} layout {s.launch_info()['layout']} {t:8.3f} ms {batch*(50 if nx>64 else 100)/t/1e3:8.1f} M iters/s", flush=True)
        s.reset()
