"""MI355X-native (gfx950) tree-attention hot path of DynamicTreeAttn behind the reference's
Python operator surface.  See DESIGN.md.  Sub-modules import lazily so that the host-side logic
(``trie``, ``synth``, ``data_parallel``) works on a box without a GPU; anything that computes
on tokens or activations requires the HIP library and raises if it is missing."""
__version__ = "0.1.0"
