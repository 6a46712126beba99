// Stand-in for <rclcpp_components/register_node_macro.hpp> — see oracle/stubs/README.md.
#pragma once
#define RCLCPP_COMPONENTS_REGISTER_NODE(NodeClass)
