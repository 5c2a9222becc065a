import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from pyqg_generative_amd.tools.simulate import generate_subgrid_forcing
from pyqg_generative_amd.tools.parameters import EDDY_PARAMS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t0 = time.time()
dealias = sys.argv[2] if len(sys.argv) > 2 else '3/2-rule'
out = generate_subgrid_forcing([64], dict(EDDY_PARAMS.nx(256), log_level=0), n_members=B, seeds=range(B), operators=('Operator1', 'Operator5', 'Operator2'), dealias=dealias)
print('seconds', time.time() - t0)
for key, ds in out.items():
    q = np.asarray(ds['q'].values).astype('float64'); f = np.asarray(ds['q_forcing_advection'].values).astype('float64')
    print(key, ds['q'].dims, q.shape, 'std q', q.std(), 'std forcing', f.std())
    # per-run values to gauge the sampling error
    qs = q.reshape(B, -1).std(1) if B > 1 else q.std(); fs = f.reshape(B, -1).std(1) if B > 1 else f.std()
    print('   per-run std q: mean %.4e sd %.2e ; forcing mean %.4e sd %.2e' % (qs.mean(), qs.std(), fs.mean(), fs.std()))
print('published (eddy/64/sharp, 300 runs): std q 5.701264812550008e-06, std forcing 4.999136229013802e-12')
