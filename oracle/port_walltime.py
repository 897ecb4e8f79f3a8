"""Wall time of the CPU port (oracle/stove_oracle.py) next to the reference itself (/root/reference imported), same inputs, same
threads, in THIS container -> tests/golden/g18_port_walltime.json (numbers only travel; bench.py quotes them in `cpu_baseline`).

TEST INFRASTRUCTURE.  BASELINE.md section 3 promises a port 'within ~10 % of the reference's wall time'; this records what it is.
Protocol: billiards, B sequences x T = 100 frames, float32, torch.set_num_threads(8) (the reference's config.max_threads, main.py:134),
Stove.forward + backward; one warm-up iteration, then the median of 3, alternating reference / port."""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as M  # noqa: E402  (imports the reference with its stubs)
import stove_oracle as O  # noqa: E402
import torch  # noqa: E402


def _cpu_model():
    with open('/proc/cpuinfo') as f:
        for line in f:
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    return '?'


def main():
    torch.set_num_threads(8)
    out = {'what': 'oracle/port_walltime.py: reference Stove.forward+backward vs the CPU port on the same inputs, this container '
                   '(8 threads, float32, median of 3 after a warm-up)', 'threads': 8, 'cpu': _cpu_model(),
           'cases': {}}
    for B in (2, 8):
        T = 100
        c = M.ref_config(torch.float32, num_obj=3)
        st = M.Stove(c)
        M.fill(st, '', 'init')
        x = torch.from_numpy(M.billiards_frames(B, T)).float()
        co = O.default_config(num_obj=3)
        structs = O.build_structs(co)
        params = {k: p.detach().clone().requires_grad_() for k, p in st.named_parameters()}
        g = torch.Generator().manual_seed(1)

        def ref_step():
            t0 = time.perf_counter()
            elbo, _, _ = st(x, 1, None)
            (-elbo).backward()
            dt = time.perf_counter() - t0
            st.zero_grad()
            return dt

        def port_step():
            eps = O.draw_eps(B, 3, T, generator=g)
            t0 = time.perf_counter()
            elbo, _ = O.stove_forward(co, params, structs, x, eps)
            (-elbo).backward()
            dt = time.perf_counter() - t0
            for p in params.values():
                p.grad = None
            return dt
        ref_step(), port_step()
        tr, tp = [], []
        for _ in range(3):
            tr.append(ref_step())
            tp.append(port_step())
        tr.sort(), tp.sort()
        out['cases'][f'B{B}_T{T}'] = {'reference_s': round(tr[1], 3), 'port_s': round(tp[1], 3), 'port_over_reference': round(tp[1] / tr[1], 3),
                                      'reference_frames_per_s': round(B * T / tr[1], 1), 'port_frames_per_s': round(B * T / tp[1], 1)}
        print(B, out['cases'][f'B{B}_T{T}'])
    torch.set_default_dtype(torch.float32)
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'g18_port_walltime.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
