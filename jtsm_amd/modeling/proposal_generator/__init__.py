from .build import PROPOSAL_GENERATOR_REGISTRY, build_proposal_generator
from .proposal_utils import add_ground_truth_to_proposals, find_top_rpn_proposals
from .rpn import RPN, RPN_HEAD_REGISTRY, StandardRPNHead, build_rpn_head
