"""r schedules and the throughput helper (reference: tome/utils.py:15-108)."""
from __future__ import annotations

import time
from typing import List, Tuple, Union

import torch


def parse_r(num_layers: int, r: Union[List[int], Tuple[int, float], int]) -> List[int]:
    """Turn ``r`` into one value per layer (tome/utils.py:83-108).

    int -> constant; ``(r, inflect)`` -> linear ramp whose mean is r, ``inflect`` in [-1, 1] giving
    the trend (-1 decreasing, 0 constant, +1 increasing); list -> taken as is, zero padded.
    """
    if isinstance(r, list):
        return list(r) + [0] * max(0, num_layers - len(r))
    inflect = 0
    if isinstance(r, tuple):
        r, inflect = r
    lo = int(r * (1.0 - inflect))
    hi = 2 * r - lo
    slope = (hi - lo) / (num_layers - 1)
    return [int(lo + slope * layer) for layer in range(num_layers)]


def benchmark(
    model: torch.nn.Module,
    device: torch.device = 0,
    input_size: Tuple[int] = (3, 224, 224),
    batch_size: int = 64,
    runs: int = 40,
    throw_out: float = 0.25,
    use_fp16: bool = False,
    verbose: bool = False,
) -> float:
    """Images (or frames, for 4-D ``input_size``) per second on random inputs, first ``throw_out``
    fraction of the runs discarded as warm-up (tome/utils.py:15-80)."""
    if not isinstance(device, torch.device):
        device = torch.device(device)
    on_gpu = device.type == "cuda"
    model = model.eval().to(device)
    batch = torch.rand(batch_size, *input_size, device=device)
    if use_fp16:
        batch = batch.half()
    per_run = batch_size if len(input_size) == 3 else batch_size * input_size[1]
    warm = int(runs * throw_out)
    done = 0
    t0 = time.time()
    with torch.autocast(device.type, enabled=use_fp16), torch.no_grad():
        for it in range(runs):
            if it == warm:
                if on_gpu:
                    torch.cuda.synchronize()
                done = 0
                t0 = time.time()
            model(batch)
            done += per_run
    if on_gpu:
        torch.cuda.synchronize()
    rate = done / (time.time() - t0)
    if verbose:
        print(f"Throughput: {rate:.2f} im/s")
    return rate
