import sys, numpy as np
sys.path.insert(0, ".")
from tests.support.problems import abi_args
from tests.test_oracle_golden import load_qp
from pmpc_amd import backend
name = sys.argv[1]
args, kw, Nc, Xg, Ug, _ = load_qp(name)
X, U = backend.lqp_solve(*abi_args(args, kw, Nc), verbose=True)
print("err", np.linalg.norm(X - Xg) / np.linalg.norm(Xg))
