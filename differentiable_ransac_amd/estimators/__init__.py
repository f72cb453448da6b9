from .essential_matrix_estimator_nister import EssentialMatrixEstimatorNister  # noqa: F401
from .essential_matrix_estimator_stewenius import EssentialMatrixEstimator  # noqa: F401
from .fundamental_matrix_estimator import FundamentalMatrixEstimatorNew, FundamentalMatrixEstimator  # noqa: F401
from .rigid_transformation_SVD_based_solver import RigidTransformationSVDBasedSolver  # noqa: F401
