"""Kernel-name fragments -> the launch label prefix ops.py gives the launch (profiling.span).  One label may cover several
dispatches (stem_mark+scan = two kernels).  Shared by pmc_summary.py and label_durations.py."""
KIND = (
    (("conv_igemm", "conv_rows", "conv_wino"), "conv_cl[", 1),
    (("gather_scatter_cl",), "gather_scatter_cl[", 1),
    (("pointnet_scatter",), "pointnet_scatter[", 1),
    (("point_head",), "point_head[", 1),
    (("stem_gemm",), "stem_gemm[", 1),
    (("stem_epilogue",), "stem_epilogue[", 1),
    (("upconv_xpass",), "upconv_xpass[", 1),
    (("upconv_ypass", "upconv_fused"), "upconv_ypass[", 1),
    (("upconv_xy",), "upconv_xy[", 1),
    (("msda_fwd",), "msda_fwd[", 1),
    (("pool_branch",), "downsample_pool_branch[", 1),
    (("tfusion_layer",), "tfusion_layer[", 1),
    (("tfusion_project",), "tfusion_project[", 1),
    (("stem_mark", "stem_scan"), "stem_mark+scan[", 2),
)


def steady_step(rows, key):
    """The dispatches of one steady-state step: between the last two tta_argmax launches (rows sorted by `key`)."""
    rows = sorted(rows, key=key)
    marks = [i for i, r in enumerate(rows) if "tta_argmax" in r["Kernel_Name"]]
    if len(marks) < 2:
        return rows
    return rows[marks[-2] + 1:marks[-1] + 1]


def match(step, labels):
    """[(label or None, [dispatch rows])] for the step: kind by kind and in order, labels of a kind are paired with the
    dispatches of that kind; everything else comes back unlabelled.  Returns (pairs, problems)."""
    used, pairs, problems = set(), [], []
    for frags, prefix, per in KIND:
        want = [l for l in labels if l.startswith(prefix)]
        have = [i for i, r in enumerate(step) if any(f in r["Kernel_Name"] for f in frags)]
        if len(want) * per != len(have):
            if want or have:
                problems.append("%d dispatches of %s vs %d labels x %d" % (len(have), "/".join(frags), len(want), per))
            continue
        for k, l in enumerate(want):
            idx = have[k * per:(k + 1) * per]
            used.update(idx)
            pairs.append((l, [step[i] for i in idx]))
    for i, r in enumerate(step):
        if i not in used:
            pairs.append((None, [r]))
    return pairs, problems
