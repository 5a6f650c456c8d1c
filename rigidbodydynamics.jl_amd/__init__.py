"""rigidbodydynamics.jl_amd — MI355X-native batched rigid-body dynamics behind RigidBodyDynamics.jl's
`dynamics!` / `inverse_dynamics!` / `mass_matrix!` / `dynamics_bias!` (src/mechanism_algorithms.jl).

Host side (this package, Python because Julia is not available in the build image): model description +
flattening (mechanism.py, urdf.py, builders.py) and the batched mirror of the reference's operator
interface over the C ABI of csrc/librbd_hip.so (state.py, _capi.py).  All arithmetic per state runs in the
hand-written HIP kernels; there is no CPU fallback."""
from .mechanism import (DEFAULT_GRAVITATIONAL_ACCELERATION, CartesianFrame3D, Fixed, FlatModel, Joint, JointType, Mechanism,
                        Planar, Prismatic, QuaternionFloating, QuaternionSpherical, Revolute, RigidBody, SinCosRevolute,
                        SpatialInertia, Transform3D, attach_, flatten, rand_configuration, rand_velocity,
                        remove_fixed_tree_joints_, rot_z_y_x, rotation_between, ContactPoint, HalfSpace3D, HuntCrossleyModel, SoftContactModel,
                        ViscoelasticCoulombModel, add_contact_point_, add_environment_primitive_, hunt_crossley_hertz)
from .urdf import default_urdf_joint_types, parse_pose, parse_urdf, write_urdf
from .builders import (maximal_coordinates, FOUR_BAR_INITIAL_Q, FOUR_BAR_INITIAL_V, double_pendulum, four_bar_linkage, quickstart_double_pendulum,
                       rand_tree_mechanism, randmech, tree_mechanism)
from .flatio import load_flat_model, save_flat_model
from . import _capi
from .state import (PDGains, SE3PDGains, default_constraint_stabilization_gains, TorqueTable, PDControl, jit_source, jit_precompile, jit_status, bank_plan, track_plan, reroot_plan, momentum, momentum_rate_bias, geometric_jacobian_, chain_plan, center_of_mass, gravitational_potential_energy, kinetic_energy, momentum_matrix_, DimensionMismatch, DynamicsResult, MechanismState, dynamics_, dynamics_bias_, dynamics_ode_, inverse_dynamics_, mass_matrix_, unpack_lower,
                    mass_matrix_solve_, rand_, set_configuration_, set_velocity_, simulate_, sync, last_kernel, zero_configuration_)
from .distributed import Comm, gather_results, shard_range, shard_sizes
