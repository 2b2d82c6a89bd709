"""skyeye.core -- model package of the MI355X-native SkyEye engine."""
