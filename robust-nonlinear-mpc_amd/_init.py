"""MI355X-native batched fast-SLS QP path.  Public names re-exported here."""
from .models import ModelData, pendulum, quadrotor, rocket, get_model  # noqa: F401
from .fast_sls import BatchedFastSLS, fast_SLS  # noqa: F401,E402
from .synthetic import make_batch  # noqa: F401,E402
from .closed_loop import ClosedLoopMPC  # noqa: F401,E402
from .monte_carlo import run_monte_carlo, disturbance_stream  # noqa: F401,E402
