"""The whole Cube R-CNN training step over the other bottom-ups of the reference's configs (DenseNet-121, MNASNet-1.0,
ShuffleNet-V2) against fixtures written by the REFERENCE's own RCNN3D (oracle/make_golden.py --backbones; its backbone wrappers
run over the oracle's restatement of the torchvision models): anchor labels and sampled ROIs exactly, the ten losses within 1e-4,
every parameter gradient within the caps of tests/test_model_parity.py.  (Named to run last: the kernels underneath are already
covered per backbone by tests/test_{densenet,mnasnet,shufflenet}_backbone.py.)"""
import os

import pytest

from test_model_parity import _run

# (MNASNet at 2 x 128 x 128 like on the GPU: at 1 x 64 x 64 its stem BatchNorm gradient sits at 3.3 % under the emulator since the round-3
# change of the Winograd interpolation points moved the summation order -- the same conditioning argument as for the GPU list below)
FIXTURES = ["densenet_tiny", "mnasnet_small", "shufflenet_tiny"]
# heads / FPN gradients: 2 % here instead of the 1 % of the DLA / ResNet fixtures -- the 24-channel p2 of MNASNet / ShuffleNet at
# 16 x 16 makes fpn_output2's weight gradient (0.02 in norm) the worst-conditioned tensor: 1.2 % on its largest elements under the
# host emulator, with every loss at 1e-4 and every gradient NORM inside the caps
HEAD_CAP = 2e-2


@pytest.mark.skipif(os.environ.get("OMNI_SLOW") != "1", reason="minutes each under the host emulator; set OMNI_SLOW=1 (the GPU variant is the gate)")
@pytest.mark.parametrize("name", FIXTURES)
def test_training_step_other_backbones_emulated(emu_lib, name):
    _run("cpu", name, head_cap=HEAD_CAP)


# On the GPU: DenseNet at the tiny size; MNASNet / ShuffleNet at 2 x 128 x 128 -- at 1 x 64 x 64 their deepest BatchNorms see 4
# samples per channel behind chains of depthwise convolutions, and the GPU's summation order then moves a logged scalar and one
# gradient norm past the 1e-4 / 3 % bars.
GPU_FIXTURES = ["densenet_tiny", "mnasnet_small", "shufflenet_small"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_FIXTURES)
def test_training_step_other_backbones_gpu(hip_lib, name):
    _run("cuda", name, head_cap=HEAD_CAP)
