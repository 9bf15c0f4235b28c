"""Golden vectors for the actor forward pass, from the reference's OWN modules
(/root/reference/robotarium_gym/utilities/rnn_agent.py, rnn_ns_agent.py), run HERE only.
Small random weights (not the model zoo's: those stay in the reference) -> tests/golden/actor_*.npz."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/robotarium_gym/utilities"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    pkg = types.ModuleType("robotarium_gym")
    util = types.ModuleType("robotarium_gym.utilities")
    sys.modules.update({"robotarium_gym": pkg, "robotarium_gym.utilities": util})
    ra = _load("robotarium_gym.utilities.rnn_agent", os.path.join(REF, "rnn_agent.py"))
    rns = _load("robotarium_gym.utilities.rnn_ns_agent", os.path.join(REF, "rnn_ns_agent.py"))
    # H = 16 keeps the fixtures small; the *_h64 ones have the hidden size of the model zoo's
    # non-shared actors (the fused MFMA kernel of csrc/actor_mfma.hip takes 64 or 128)
    for name, cls, use_rnn, H in (("actor_shared_gru", ra.RNNAgent, True, 16), ("actor_shared_mlp", ra.RNNAgent, False, 16),
                                  ("actor_ns_gru", rns.RNNNSAgent, True, 16),
                                  ("actor_shared_gru_h64", ra.RNNAgent, True, 64),
                                  ("actor_ns_gru_h64", rns.RNNNSAgent, True, 64)):
        torch.manual_seed(7)
        N, I, A, T = 4, 20, 5, 6
        args = types.SimpleNamespace(hidden_dim=H, n_actions=A, use_rnn=use_rnn, n_agents=N)
        model = cls(I, args)
        x = torch.randn(T, N, I)
        hs = torch.zeros(N, H) if cls is ra.RNNAgent else torch.zeros(1, N, H)
        qs, hh = [], []
        with torch.no_grad():
            for t in range(T):
                if cls is ra.RNNAgent:
                    q, hs = model(x[t], hs)                       # misc.py:167-168
                else:
                    q, hs = model(x[t], hs)                       # misc.py:165-166 (NS: hidden [1,N,H])
                qs.append(q.reshape(N, A).numpy().copy())
                hh.append(hs.reshape(N, H).numpy().copy())
        d = {"inputs": x.numpy(), "q": np.stack(qs), "h": np.stack(hh), "use_rnn": np.array(use_rnn)}
        for k, v in model.state_dict().items():
            d["sd_" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, d["q"].shape)


if __name__ == "__main__":
    main()
