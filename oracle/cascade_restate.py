"""TEST ORACLE — NOT PRODUCT CODE.

numpy restatement of the reference's cascade stage loop around the hot call (all [R]: the code is in /root/reference):

    stage loop                                   FaceDetectUpdated.py:665-766
    update_current_subimage_coordinates          face_analysis.py:803-840
    identify_patches_to_discard                  face_analysis.py:842-887
    per-level constants                          FaceDetectUpdated.py:595-605, face_analysis.py:651-652

The three third-party calls of a stage (sub-image extraction, flow.execute, classifier.regression) are injected, so the
same loop can be driven by PIL + oracle flows (pure CPU) or by the product's own features when only the glue is under test.
Candidates of all pyramid levels run as one batch (the reference's comment at FaceDetectUpdated.py:599 — rows never
interact, so this only changes the order of the survivors), each carrying the constants of its own level.
"""
from __future__ import annotations

import numpy as np

CUT_OFFS_FACE = [0.99, 0.95, 0.85, 0.8, 0.7, 0.6, 0.5, 0.45, 0.10, 0.05]        # FaceDetectUpdated.py:98
TOLERANCE_SCALE, TOLERANCE_ANGLE, TOLERANCE_POSXY = 1.1, 1.1, 1.1              # FaceDetectUpdated.py:113-115


def level_constants(sampling_value, subimage_width, subimage_height, regression_width, regression_height, net_Dx, net_Dy):
    """(max_Dx_diff, max_Dy_diff, base_side) of one pyramid level (face_analysis.py:651-652; FaceDetectUpdated.py:604-605)."""
    pw, ph = subimage_width * sampling_value, subimage_height * sampling_value
    return net_Dx * pw / regression_width, net_Dy * ph / regression_height, np.sqrt(pw ** 2 + ph ** 2)


def update_coordinates(network_type, coords, angles, reg_out, regression_width, regression_height, desired_sampling=0.825):
    """face_analysis.py:803-840 (in place on copies the caller owns)."""
    if network_type == "Disc":
        pass
    elif network_type == "PosX":
        width = coords[:, 2] - coords[:, 0]
        r = reg_out * width / regression_width
        coords[:, 0] = coords[:, 0] - r
        coords[:, 2] = coords[:, 2] - r
    elif network_type == "PosY":
        height = coords[:, 3] - coords[:, 1]
        r = reg_out * height / regression_height
        coords[:, 1] = coords[:, 1] - r
        coords[:, 3] = coords[:, 3] - r
    elif network_type == "PAng":
        angles = angles + reg_out
    elif network_type == "Scale":
        old_width = coords[:, 2] - coords[:, 0]
        old_height = coords[:, 3] - coords[:, 1]
        x_center = (coords[:, 2] + coords[:, 0]) / 2.0
        y_center = (coords[:, 3] + coords[:, 1]) / 2.0
        width = old_width / reg_out * desired_sampling
        height = old_height / reg_out * desired_sampling
        coords[:, 0] = x_center - width / 2.0
        coords[:, 2] = x_center + width / 2.0
        coords[:, 1] = y_center - height / 2.0
        coords[:, 3] = y_center + height / 2.0
    else:
        raise Exception("Network type unknown!!!: %s" % network_type)
    return coords, angles


def patches_to_discard(network_type, coords, angles, reg_out, orig_index, orig_coords, orig_angles, orig_level, net_mins, net_maxs,
                       net_Dang, cut_off_face):
    """face_analysis.py:842-887 with max_Dx_diff / max_Dy_diff / base_side taken per original window."""
    oc = orig_coords[orig_index]
    if network_type == "PosX":
        d = (coords[:, 2] + coords[:, 0]) / 2 - (oc[:, 2] + oc[:, 0]) / 2
        return np.abs(d) > (orig_level[orig_index, 0] * TOLERANCE_POSXY)
    if network_type == "PosY":
        d = (coords[:, 3] + coords[:, 1]) / 2 - (oc[:, 3] + oc[:, 1]) / 2
        return np.abs(d) > (orig_level[orig_index, 1] * TOLERANCE_POSXY)
    if network_type == "PAng":
        oa = orig_angles[orig_index]
        return (angles > oa + net_Dang * TOLERANCE_ANGLE) | (angles < oa - net_Dang * TOLERANCE_ANGLE)
    if network_type == "Scale":
        sides = np.sqrt(((coords[:, 0:2] - coords[:, 2:4]) ** 2).sum(axis=1))
        ratio = sides / orig_level[orig_index, 2]
        return (ratio > (net_maxs / 0.825) * TOLERANCE_SCALE) | (ratio < (net_mins / 0.825) / TOLERANCE_SCALE)
    if network_type == "Disc":
        return reg_out >= cut_off_face
    raise Exception("Unknown network type:" + str(network_type))


def run_cascade(stage_types, has_network, orig_coords, orig_level, pipeline, extract, execute, regress):
    """The stage loop (FaceDetectUpdated.py:665-766) over one batch of original windows.

    stage_types: ["Disc1", "PosX0", ...] (type + serial digit, :669-670); has_network[k]: networks[k] is not None.
    extract(coords, delta_angs) -> (n, w*h); execute(k, subimages) -> sl; regress(k, sl) -> reg_out.
    Returns dict(coords, angles, orig_index, confidence, counts=[survivors after every stage], rows_executed)."""
    coords, angles = orig_coords.copy(), np.zeros(len(orig_coords))
    orig_angles = np.zeros(len(orig_coords))
    orig_index = np.arange(len(orig_coords))
    conf = np.zeros(len(orig_coords))
    subs = sl = None
    counts, rows_executed = [], 0
    for k, st in enumerate(stage_types):
        ntype, serial = st[:-1], int(st[-1])
        skip_extract = (k > 0 and stage_types[k - 1][:-1] == "Disc") or not has_network[k]          # :674-681
        if not skip_extract:
            subs = extract(coords, -1 * angles)                                                      # :686, face_analysis.py:782
        if len(coords) > 0:
            if has_network[k]:
                sl = execute(k, subs)                                                                # :699
                rows_executed += len(subs)
            reg = regress(k, sl) if len(sl) else np.zeros(0)                                         # :719
            coords, angles = update_coordinates(ntype, coords, angles, reg, pipeline["regression_width"], pipeline["regression_height"])
            wrong = patches_to_discard(ntype, coords, angles, reg, orig_index, orig_coords, orig_angles, orig_level,
                                       pipeline["net_mins"], pipeline["net_maxs"], pipeline["net_Dang"], CUT_OFFS_FACE[serial])
            keep = ~wrong                                                                            # :739-759
            coords, angles, orig_index = coords[keep].copy(), angles[keep].copy(), orig_index[keep].copy()
            sl, subs = sl[keep].copy(), subs[keep].copy()
            conf = reg[keep].copy() if ntype == "Disc" else conf[keep].copy()
        counts.append(len(coords))
    return dict(coords=coords, angles=angles, orig_index=orig_index, confidence=conf, counts=counts, rows_executed=rows_executed)
