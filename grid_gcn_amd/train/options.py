"""Path switches of the training / evaluation host code: ONE object, `OPT`.

Every switch selects between two implementations of the same step that compute the same thing (the defaults are
what is measured and shipped); they exist for A/B measurements (`bench.py --switch NAME=0`) and for the tests that
run one input through both paths.  Nothing here, and nothing in the C library, reads the process environment.

    from grid_gcn_amd.train.options import OPT
    with OPT.override(NOZ_ATT_BWD=False):      # scoped (tests)
        ...
    OPT.set("FOLD_FINALIZE", False)            # process-wide (bench.py --switch)

Kernel-selection options of the C library itself (gridgcn_set_option: COL_SPLIT, ATT_NZ_V2, BWD_FUSED128, ...) are a
separate, process-wide table inside the library; bench.py's --switch takes both kinds of name.
(Until round 5 these were ~20 module-level booleans of a 2 570-line train_ops.py that bench.py mutated with setattr.)
"""
import contextlib
import dataclasses


@dataclasses.dataclass
class PathOptions:
    # register-direct forward / dX kernels (csrc/gridgcn_direct.hip) for row widths that are a multiple of 8
    DIRECT_FWD: bool = True
    DIRECT_DX: bool = True
    # first conv of the point MLP applied to the source points and gathered (csrc/gridgcn_edgelin.hip)
    SRC_FIRST_CONV: bool = True
    # ... and, for single-layer point MLPs, recomputed by its consumers instead of stored
    NO_Z0: bool = True
    # ... and its backward reduced to the sparse arg-max entries (gg_k_edge_lin0_bwd_sparse)
    SPARSE_L0: bool = True
    # bf16 mode (set_mlp_precision("bf16")): bf16 STORAGE of the attention pre-activation of the up layers
    Z16_STORAGE: bool = True
    # fp32 mode, up layers: backward of the second attention conv without its [E, 128] pre-activation
    # (csrc/gridgcn_attbwd_nz.hip): the tensor is not kept for the backward at all
    NOZ_ATT_BWD: bool = True
    # ... and the forward: its BatchNorm from the moments of the 32-wide activation, the conv recomputed inside the
    # pair product / max kernel (csrc/gridgcn_attfwd.hip): the tensor is never written (needs NOZ_ATT_BWD)
    NOZ_ATT_FWD: bool = True
    # ... and the backward takes S1 = sum a1, S2 = sum a1 a1^T from the moments that forward left behind instead of
    # accumulating them again (gridgcn_att_bwd_noz_mom: 144 MFMAs per tile instead of 160, one launch fewer).
    # ROUND 6, WRITTEN WHILE THE GPU POOL WAS CLOSED: off until a GPU session has run tests/test_zz_r6_unverified.py
    NOZ_BWD_MOMENTS: bool = False
    # bf16 mode: the same pair of (fp32) kernels instead of the bf16-stored tensor, where the shape allows
    NOZ_IN_BF16: bool = True
    # the source-point products on csrc/gridgcn_gemm.hip instead of the framework's GEMM
    SMALL_GEMM: bool = True
    # small zero-filled accumulators carved from 4 MB zero chunks (one fill per chunk instead of ~70 per step)
    ZERO_ARENA: bool = True
    # the optimizer of bench.py / the tests' training loops: grid_gcn_amd.optim.Adam (one launch)
    OWN_ADAM: bool = True
    # BatchNorm finalisation of a conv layer by the forward kernel's last workgroup instead of a launch of its own
    FOLD_FINALIZE: bool = True
    # BatchNorm statistics of a single-layer point MLP from per-source counts and geo_vec sums ...
    SRC_STATS: bool = True
    # ... from this many edges on (below: the edge pass is a 10-20 us launch, these are two)
    SRC_STATS_MIN_EDGES: int = 1 << 19
    # head: Dropout evaluated inside fc2's forward / dW kernels (no dropped tensor)
    FUSE_DROPOUT: bool = True
    # geo_vec weight + bias table of the source-side first conv built by the prepack launch
    WGB_PREPACK: bool = True
    # concat + centre mask + zero padding of a layer boundary in one launch (model.GGCNSeg.forward)
    GLUE_KERNELS: bool = True
    # training: the index operators behind the first Gridify on a side stream (model._IndexAhead).  ROUND 6, WRITTEN
    # WHILE THE GPU POOL WAS CLOSED: off until a GPU session has run tests/test_zz_r6_unverified.py and an A/B
    INDEX_SIDE_STREAM: bool = False
    # evaluation of single-layer-pt edge blocks (the up layers) through the source-side kernels ...
    SRC_EVAL: bool = True
    # ... with the second attention conv + product + max in one kernel (csrc/gridgcn_atteval.hip)
    ATT_MAX_EVAL: bool = True
    # folded BatchNorm vectors / packed weights of evaluation cached per module (train/evalpath.py)
    EVAL_CACHE: bool = True
    # train.common.LaunchTimers or None: device time of selected library calls inside eager steps (bench.py)
    TIMERS: object = None

    def set(self, name, value):
        """bench.py --switch NAME=0|1 (booleans) / NAME=<int>; raises on an unknown name"""
        f = {x.name: x for x in dataclasses.fields(self)}.get(name)
        if f is None or name == "TIMERS":
            raise KeyError("no path switch %r (have: %s)" % (name, ", ".join(self.names())))
        cur = getattr(self, name)
        setattr(self, name, bool(int(value)) if isinstance(cur, bool) else int(value))

    def names(self):
        return [x.name for x in dataclasses.fields(self) if x.name != "TIMERS"]

    @contextlib.contextmanager
    def override(self, **kw):
        old = {k: getattr(self, k) for k in kw}       # (AttributeError on an unknown name)
        try:
            for k, v in kw.items():
                setattr(self, k, v)
            yield self
        finally:
            for k, v in old.items():
                setattr(self, k, v)


OPT = PathOptions()
