"""The stand-alone mirror of the reference's public surface (csrc/host/nos_reference_api.hpp) must keep agreeing with
the reference's own headers — member names, types, default values, enumerators, constructor / Evaluate / Solve
signatures, the loss functions' arithmetic — because the drop-in classes compile against either one
(-DNOS_IN_REFERENCE_TREE) and the in-tree build cannot be exercised here (no Eigen).

Runs in the build container only: the reference is not mounted on the GPU box (skipped there); nothing of the reference
travels — the headers are parsed where they lie."""
import os
import re

import pytest

REF = "/root/reference/nonlinear_optimizer"
MIRROR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nonlinear_optimizer_for_slam_amd", "csrc",
                      "host", "nos_reference_api.hpp")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")


def _clean(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text)


def _block(text, header_regex):
    """Body of the first `{ ... }` that follows header_regex (brace matched)."""
    m = re.search(header_regex, text)
    assert m, header_regex
    i = text.index("{", m.end() - 1)
    depth, j = 0, i
    while True:
        depth += text[j] == "{"
        depth -= text[j] == "}"
        if depth == 0:
            return text[i + 1:j]
        j += 1


def _members(body):
    """[(type, name, default)] of `type name{default};` members, nested anonymous structs flattened as name.member."""
    out = []
    rest = body
    for m in re.finditer(r"struct\s*\{", body):
        inner = _block(body[m.start():], r"struct\s*")
        tail = body[m.start() + body[m.start():].index(inner) + len(inner) + 1:]
        name = re.match(r"\s*(\w+)\s*;", tail).group(1)
        out += [(t, name + "." + n, d) for t, n, d in _members(inner)]
        rest = rest.replace("struct {" + inner + "} " + name + ";", " ").replace("struct {" + inner + "}" + name + ";", " ")
    # drop member functions (anything with a parameter list before a body) before looking for data members
    rest = re.sub(r"[\w:<>~&\*\s]+\([^()]*\)\s*(const)?\s*(final|override)?\s*(:[^{]*)?\{[^{}]*(\{[^{}]*\}[^{}]*)*\}", " ", rest)
    for m in re.finditer(r"((?:const\s+)?[\w:]+(?:<[^<>]*>)?)\s+(\w+)\s*\{([^{}]*)\}\s*;", rest):
        out.append((m.group(1).replace(" ", ""), m.group(2), m.group(3).replace(" ", "")))
    return out


def _enum(text, name):
    return [e.strip().replace(" ", "") for e in _block(text, r"enum class " + name + r"\s*").split(",") if e.strip()]


def _signature(text, pattern):
    m = re.search(pattern, text)
    assert m, pattern
    return re.sub(r"\s+", " ", m.group(0)).replace("( ", "(").replace(" )", ")").strip()


@pytest.fixture(scope="module")
def mirror():
    return _clean(MIRROR)


def test_options_and_enums(mirror):
    ref = _clean(os.path.join(REF, "options.h"))
    assert _members(_block(ref, r"struct Options\s*")) == _members(_block(mirror, r"struct Options\s*"))
    assert len(_members(_block(ref, r"struct Options\s*"))) == 8  # 3 + 3 + 2: the parser saw all of them
    for e in ("MinimizerType", "LinearSolverType"):
        assert _enum(ref, e) == _enum(mirror, e)


def test_ndt_and_correspondence_records(mirror):
    mdm = _clean(os.path.join(REF, "mahalanobis_distance_minimizer", "types.h"))
    rem = _clean(os.path.join(REF, "reprojection_error_minimizer", "types.h"))
    mirror_mdm = _block(mirror, r"namespace mahalanobis_distance_minimizer\s*")
    mirror_rem = _block(mirror, r"namespace reprojection_error_minimizer\s*")
    want = _members(_block(mdm, r"struct NDT\s*"))
    assert [n for _, n, _ in want] == ["count", "sum", "moment", "mean", "information", "sqrt_information", "is_valid", "is_planar"]
    assert want == _members(_block(mirror_mdm, r"struct NDT\s*"))
    # `NDT ndt;` has no initialiser: compare the whole member list textually
    ref_c = re.sub(r"\s+", "", _block(mdm, r"struct Correspondence\s*"))
    assert ref_c == re.sub(r"\s+", "", _block(mirror_mdm, r"struct Correspondence\s*")) == "Vec3point{Vec3::Zero()};NDTndt;"
    assert _members(_block(rem, r"struct CameraIntrinsics\s*")) == _members(_block(mirror_rem, r"struct CameraIntrinsics\s*"))
    assert len(_members(_block(rem, r"struct CameraIntrinsics\s*"))) == 8
    assert _members(_block(rem, r"struct Correspondence\s*")) == _members(_block(mirror_rem, r"struct Correspondence\s*"))


def test_loss_functions(mirror):
    ref = _clean(os.path.join(REF, "loss_function.h"))
    for cls in ("ExponentialLossFunction", "HuberLossFunction"):
        rb, mb = _block(ref, r"class " + cls + r"\s*:\s*public LossFunction\s*"), _block(mirror, r"class " + cls + r"\s*:\s*public LossFunction\s*")
        # constructor parameter list
        rc = re.search(cls + r"\(([^()]*)\)", rb).group(1)
        mc = re.search(cls + r"\(([^()]*)\)", mb).group(1)
        assert re.sub(r"\s+", " ", rc) == re.sub(r"\s+", " ", mc)
        # the scalar Evaluate body, statement for statement
        r_eval = _block(rb, r"void Evaluate\(const double squared_residual, double output\[\d\]\) final\s*")
        m_eval = _block(mb, r"void Evaluate\(const double squared_residual, double output\[\d\]\) final\s*")
        assert re.sub(r"\s+", "", r_eval) == re.sub(r"\s+", "", m_eval), cls
        # argument checks: same conditions, same exception type and message
        assert re.findall(r"if \(([^()]*)\)\s*throw (std::\w+)\(\"([^\"]*)\"\)", rb) == \
            re.findall(r"if \(([^()]*)\)\s*throw (std::\w+)\(\"([^\"]*)\"\)", mb)
        # private data members
        assert _members(rb.split("private:")[1]) == _members(mb.split("private:")[1])
    assert "virtual void Evaluate(const double squared_residual, double* output) = 0;" in ref
    assert "virtual void Evaluate(const double squared_residual, double* output) = 0;" in mirror


def test_solver_base_classes(mirror):
    mdm = _clean(os.path.join(REF, "mahalanobis_distance_minimizer", "mahalanobis_distance_minimizer.h"))
    rem = _clean(os.path.join(REF, "reprojection_error_minimizer", "reprojection_error_minimizer.h"))
    mirror_mdm = _block(mirror, r"namespace mahalanobis_distance_minimizer\s*")
    mirror_rem = _block(mirror, r"namespace reprojection_error_minimizer\s*")
    solve = r"virtual bool Solve\([^()]*\) = 0;"
    assert _signature(mdm, solve) == _signature(mirror_mdm, solve) == \
        "virtual bool Solve(const Options& options, const std::vector<Correspondence>& correspondences, Pose* pose) = 0;"
    assert _signature(rem, solve) == _signature(mirror_rem, solve)
    assert "const CameraIntrinsics& camera_intrinsics, Pose* pose" in _signature(rem, solve)
    for text in (mdm, mirror_mdm, rem, mirror_rem):
        assert re.search(r"void SetLossFunction\(const std::shared_ptr<LossFunction>& loss_function\)", text)
    for text in (mdm, mirror_mdm):
        assert re.search(r"void SetMultiThreadExecutor\( ?const std::shared_ptr<MultiThreadExecutor>& multi_thread_executor\)", text)
        assert "std::shared_ptr<LossFunction> loss_function_{nullptr};" in text
        assert "std::shared_ptr<MultiThreadExecutor> multi_thread_executor_{nullptr};" in text
    assert "SetMultiThreadExecutor" not in rem and "SetMultiThreadExecutor" not in mirror_rem  # the reference has none there


def test_pose_graph_records(mirror):
    types = _clean(os.path.join(REF, "pose_graph_optimizer", "types.h"))
    pgo = _clean(os.path.join(REF, "pose_graph_optimizer", "pose_graph_optimizer.h"))
    mirror_pgo = _block(mirror, r"namespace pose_graph_optimizer\s*")
    assert _enum(types, "ConstraintType") == _enum(mirror_pgo, "ConstraintType")
    assert _members(_block(types, r"struct Constraint\s*")) == _members(_block(mirror_pgo, r"struct Constraint\s*"))
    assert re.sub(r"\s+", "", _block(pgo, r"struct PoseParameter\s*")) == re.sub(r"\s+", "", _block(mirror_pgo, r"struct PoseParameter\s*"))
