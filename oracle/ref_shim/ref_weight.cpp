// ref_weight.cpp -- harness around the TEXT of Model::computeFusionWeight and Model::rodrigues2 (Core/Model/Model.cpp:391-406, 817-865),
// cut out of the reference at build time by build_ref.py and compiled in front of this file.  TEST INFRASTRUCTURE ONLY.
//
// What the two functions need of the Model class is declared in the prologue build_ref.py writes (class WeightPinModel: getPose(),
// lastPose, getLastTransform() as Model.h:216 states it).  Eigen is not in the image: eigen_fixed supplies the fixed-size matrices with
// stated conventions, and Eigen::JacobiSVD is stood in for by the convention our restatements state as well -- for a product of rotation
// matrices the re-orthonormalisation U V^T is the matrix itself up to rounding, so the stand-in returns U = matrix, V = identity.  The pin
// therefore covers the arithmetic AROUND the SVD (angle, axis, small-angle branches, clamps, the weight formula) exactly as the reference
// spells it, and says nothing about Eigen's SVD.
extern "C" float ref_fusion_weight(const float pose[16], const float lastPose[16], float weightMultiplier)
{
    WeightPinModel m;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { m.pose(i, j) = pose[i * 4 + j]; m.lastPose(i, j) = lastPose[i * 4 + j]; }
    return m.computeFusionWeight(weightMultiplier);
}
