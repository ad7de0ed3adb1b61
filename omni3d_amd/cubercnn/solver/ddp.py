"""Owning the gradient exchange under the training script's own DistributedDataParallel wrapper.

The reference wraps the model itself (tools/train_net.py:449-454: `DistributedDataParallel(model, device_ids=[...],
broadcast_buffers=False, find_unused_parameters=True)`, the class imported straight from torch) and only then builds the optimizer
inside do_train.  DDP's reducer would all-reduce 230 parameter tensors in ~25 MB buckets from autograd hooks -- which the HIP path
does not have: weight gradients are accumulated by the backward kernels straight into the optimizer's flat bucket and never pass
through autograd, and the whole step may be one hipGraph replay.  Round 2 answered with "no direct accumulation under DDP, DDP
reduces, and the optimizer all-reduces the bucket once more to be safe" (communication doubled).  Now:

  * `build_model` (when a process group with more than one rank exists) calls `prepare_for_ddp`: every real parameter and buffer is
    named in `model._ddp_params_and_buffers_to_ignore`, which torch's DDP honours -- no initial broadcast, no hooks, no buckets for
    them -- and ONE extra 1-element parameter, `_omni_ddp_anchor`, stays visible so the wrapper has something to manage (DDP
    refuses a module without trainable parameters).  It is hidden from state dicts, never updated, and every loss dict the model
    hands to the loop depends on it with weight 0, so its hook fires once per iteration and DDP's bookkeeping stays in step;
  * `build_optimizer` sees `_omni_owns_exchange`, keeps direct accumulation, broadcasts rank 0's parameters and buffers once (what
    DDP's constructor would have done), and the flat bucket is exchanged by the optimizer's own overlapped two-range all-reduce
    (inside the staged-graph replay) or by `step()` on eager iterations -- exactly once per iteration either way."""
import torch
import torch.distributed as dist

ANCHOR = "_omni_ddp_anchor"


def world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def prepare_for_ddp(model):
    if getattr(model, "_omni_owns_exchange", False):
        return model
    dev = next(model.parameters()).device
    names = [n for n, _ in model.named_parameters()] + [n for n, _ in model.named_buffers()]
    model.register_parameter(ANCHOR, torch.nn.Parameter(torch.zeros(1, device=dev)))
    model._ddp_params_and_buffers_to_ignore = names
    model._omni_owns_exchange = True

    def drop_anchor(module, state_dict, prefix, local_metadata):
        state_dict.pop(prefix + ANCHOR, None)
        return state_dict

    def supply_anchor(state_dict, prefix, *unused):
        state_dict.setdefault(prefix + ANCHOR, torch.zeros(1))
    model._register_state_dict_hook(drop_anchor)
    model._register_load_state_dict_pre_hook(supply_anchor)
    return model


def tie_to_anchor(model, losses):
    """the loop-level loss dict depends on the anchor with weight 0: DDP's one hook fires, nothing else changes"""
    anchor = getattr(model, ANCHOR, None)
    if anchor is None or not losses:
        return losses
    k = next(iter(losses))
    losses[k] = losses[k] + 0.0 * anchor.sum()
    return losses


@torch.no_grad()
def broadcast_replica(model, optimizer):
    """rank 0's parameters (one flat bucket) and buffers to every rank: DDP's constructor did this for the parameters it manages"""
    if world() < 2:
        return
    dist.broadcast(optimizer.flat_param, src=0)
    # parameters the optimizer does not hold (frozen before build_optimizer: requires_grad False) are in DDP's ignore list too, so
    # nobody else synchronises them
    slots = getattr(optimizer, "_slot", {})
    for n, p in model.named_parameters():
        if id(p) not in slots and n != ANCHOR and p.numel():
            dist.broadcast(p.data, src=0)
    for b in model.buffers():
        if b.numel():
            dist.broadcast(b, src=0)
