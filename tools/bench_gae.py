"""Micro-benchmark of the GAE kernels on one MI355X (development tool, not the judged bench).

Prints kernel time from HIP events over back-to-back launches and the algorithmic GB/s
(17 B per env-step for the fused kernel: r 4 + v 4 + done 1 read, ret 4 + adv 4 written)."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import gae  # noqa: E402


def time_launches(fn, iters, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--horizon', type=int, default=32)
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--rotate', type=int, default=16,
                    help='number of distinct buffer sets cycled through (defeats the 256 MB MALL when large)')
    a = ap.parse_args()
    dev = 'cuda:0'
    N, H = a.envs, a.horizon
    sets = []
    for i in range(a.rotate):
        sets.append(dict(
            r=torch.randn(N, H, device=dev), v=torch.randn(N, H, device=dev),
            d=(torch.rand(N, H, device=dev) < 0.05).to(torch.uint8),
            lv=torch.randn(N, device=dev), ld=(torch.rand(N, device=dev) < 0.05).to(torch.uint8),
            ret=torch.empty(N, H, device=dev), adv=torch.empty(N, H, device=dev),
            part=torch.empty(gae.num_moment_partials(N), 6, dtype=torch.float64, device=dev)))
    k = [0]

    def fused():
        s = sets[k[0] % len(sets)]
        k[0] += 1
        gae.gae_returns_advantages(s['r'], s['v'], s['d'], s['lv'], s['ld'], 0.99, 0.95,
                                   out_returns=s['ret'], out_advantages=s['adv'],
                                   moment_partials=s['part'])

    def raw_view():
        s = sets[k[0] % len(sets)]
        k[0] += 1
        gae.compute_gae(s['r'].t().unsqueeze(2), s['v'].t().unsqueeze(2), s['d'].t(),
                        s['lv'].unsqueeze(1), s['ld'], 0.99, 0.95)

    tm = dict(r=torch.randn(H, N, 1, device=dev), v=torch.randn(H, N, 1, device=dev),
              d=(torch.rand(H, N, device=dev) < 0.05).float(), lv=torch.randn(N, 1, device=dev),
              ld=(torch.rand(N, device=dev) < 0.05).float())

    def strided_tm():
        gae.compute_gae(tm['r'], tm['v'], tm['d'], tm['lv'], tm['ld'], 0.99, 0.95)

    steps = N * H
    for name, fn, bytes_per in (('fused env-major (ret+adv+moments)', fused, 17),
                                ('env-major raw (compute_gae view)', raw_view, 13),
                                ('strided time-major f32 dones', strided_tm, 16)):
        us = time_launches(fn, a.iters)
        gbs = steps * bytes_per / us * 1e-3
        print(f'{name:40s} {us:8.2f} us/launch  {gbs:8.1f} GB/s algorithmic  '
              f'({gbs / 8000 * 100:5.1f}% of 8 TB/s)  [{steps * bytes_per / 1e6:.2f} MB]')


if __name__ == '__main__':
    main()
