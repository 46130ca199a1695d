"""Generates tests/golden/H_state_dict_keys.json: the parameter / buffer names and shapes of the reference's sub-networks
(/root/reference, imported in place with the stubs of _ref_import.py), under the attribute names PL3DVNet gives them
(mv3d/lightningmodel.py:36-43) -- the key list of a Lightning checkpoint's ``state_dict`` as far as the reference's modules
can be instantiated here: ``mvsnet.cnn_3d.*`` (CostRegNet(32, 8)), ``pointnet.*`` (PointNet(128, 64, 35)), ``decoder.net.*``
(the Conv1d stack of HypothesisDecoder(352, 128, 3, 1)), ``refine_{quarter,half,full}.*`` (PropagationNet).  Not covered:
``sparse_conv.*`` (MinkowskiEngine modules -- absent; their naming is restated in SURVEY.md 8b) and the torchvision backbone.

    python tests/golden/make_state_dict_keys.py        (build container only)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

ref = _ref_import.reference()


def keys(prefix, module):
    return {prefix + k: list(v.shape) for k, v in module.state_dict().items()}


def main():
    out = {}
    out.update(keys('mvsnet.cnn_3d.', ref.mvsnet.CostRegNet(32, 8)))
    out.update(keys('pointnet.', ref.scene.PointNet(128, 64, 35)))
    # HypothesisDecoder.__init__ builds a MinkowskiInterpolation (stubbed) next to the Conv1d stack `net`
    out.update(keys('decoder.', ref.refine.HypothesisDecoder(352, 128, 3, 1)))
    out.update(keys('refine_quarter.', ref.up.PropagationNet(33)))
    out.update(keys('refine_half.', ref.up.PropagationNet(33)))
    out.update(keys('refine_full.', ref.up.PropagationNet(4)))
    with open(os.path.join(HERE, 'H_state_dict_keys.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(len(out), 'keys')


if __name__ == '__main__':
    main()
