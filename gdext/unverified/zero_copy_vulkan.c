/*
 * zero_copy_vulkan.c -- the Vulkan half of the zero-copy path: "returns the same TextureRD" with no host hop (north star).
 *
 * COMPILE-GUARDED: neither <vulkan/vulkan.h> nor an engine exist in this image, so nothing here is built or run by this repository
 * (gdext/Makefile builds it only with CSKY_HAVE_VULKAN=1).  The HIP half it calls IS built and lives in libcloudsky.so:
 * csky_external_frame_import_fd / _import_semaphore_fd / _signal / _fence / _ready / _wait / _release (include/cloudsky.h, csrc/api.cpp);
 * its memory half is exercised on the GPU against a foreign allocator (tools/ext_frame_roundtrip.py).
 *
 * Design.  cloud_sky.gd keeps three RGBA16F textures created with rd.texture_create() (cloud_sky.gd:368-378) and hands one of them to the
 * sky material as Texture2DRD.texture_rd_rid (:137-148).  The copy path (CloudSkyHIP.collect() + rd.texture_update()) fills those textures
 * through host memory.  Here the extension creates the images ITSELF, on the engine's own VkDevice, with exportable memory:
 *
 *   1. VkDevice / VkPhysicalDevice come from RenderingDevice.get_driver_resource(DRIVER_RESOURCE_LOGICAL_DEVICE / _PHYSICAL_DEVICE, RID(), 0).
 *   2. csky_zc_create_image(): a 2-D VK_FORMAT_R16G16B16A16_SFLOAT image, VK_IMAGE_TILING_LINEAR (so that its bytes are rows of pixels a
 *      HIP kernel can address: VkSubresourceLayout.rowPitch is the row pitch handed to csky_render_clouds_device), usage SAMPLED |
 *      TRANSFER_DST, VkExternalMemoryImageCreateInfo{OPAQUE_FD}; dedicated allocation with VkExportMemoryAllocateInfo{OPAQUE_FD};
 *      vkGetMemoryFdKHR -> fd; csky_external_frame_import_fd(ctx, fd, ...) -> device pointer of the image's texels in HIP's address space.
 *   3. Ordering.  Preferred: an exportable VkSemaphore (VkExportSemaphoreCreateInfo{OPAQUE_FD}) -> vkGetSemaphoreFdKHR ->
 *      csky_external_frame_import_semaphore_fd.  MEASURED on the MI355X box: ROCm 7.2's Linux runtime answers hipErrorNotSupported to
 *      hipImportExternalSemaphore (a drm_syncobj fd, which is what an amdgpu opaque-fd semaphore is; profiles/r03/external_semaphore_probe.txt), so on this
 *      runtime the image is created WITHOUT a semaphore and ordered by the host: csky_external_frame_fence() behind the march,
 *      csky_external_frame_ready() polled at the start of the next update pass (csky_zc_ready below).
 *   4. GDScript wraps the VkImage: rid = rd.texture_create_from_extension(TEXTURE_TYPE_2D, DATA_FORMAT_R16G16B16A16_SFLOAT, TEXTURE_SAMPLES_1,
 *      TEXTURE_USAGE_SAMPLING_BIT, image_handle, w, h, 1, 1) and texture.texture_rd_rid = rid -- the SAME Texture2DRD the material already
 *      samples (cloud_sky.gd:137-148), now backed by memory the march writes.
 *   5. Per update pass: csky_render_clouds_device(ctx, pc, w, bands, zc->d_ptr, zc->row_pitch, stream), then
 *      csky_external_frame_signal(...) when a semaphore was imported (the engine's next submission that samples the texture waits on it:
 *      csky_zc_wait_semaphore() returns it for RenderingDevice's external-semaphore hook of the build in use) or csky_external_frame_fence(...)
 *      when not: the reference's three-texture ring already leaves a whole update pass between "written" and "sampled"
 *      (cloud_sky.gd:137-148), so the host rotates the ring only once csky_zc_ready() says the frame is complete -- normally true at the
 *      first poll of the next pass (a C3 frame takes 1.7 ms).
 *      The image stays in VK_IMAGE_LAYOUT_GENERAL / SHADER_READ_ONLY_OPTIMAL as the engine set it; LINEAR tiling makes the layout a no-op
 *      for the texel addresses.
 *
 * Multi-GPU: the frame lives on the device the engine renders with; csky_multi_render_clouds_device(m, ..., zc->d_ptr, ...) lets the other
 * devices store their bands into that same imported memory over xGMI (device 0 of the handle must be the engine's GPU).
 */
#ifdef CSKY_HAVE_VULKAN
#include <vulkan/vulkan.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "../../include/cloudsky.h"

typedef struct csky_zc_image {
    VkDevice device;
    VkImage image;
    VkDeviceMemory memory;
    VkSemaphore semaphore;
    VkDeviceSize allocation_bytes, row_pitch, offset;
    int width, height;
    csky_external_frame *frame;   /* the HIP view of `memory` */
    void *d_ptr;                  /* device pointer of texel (0,0) in HIP's address space */
} csky_zc_image;

static uint32_t csky_zc_memory_type(VkPhysicalDevice phys, uint32_t type_bits, VkMemoryPropertyFlags want) {
    VkPhysicalDeviceMemoryProperties mp;
    uint32_t i;
    vkGetPhysicalDeviceMemoryProperties(phys, &mp);
    for (i = 0; i < mp.memoryTypeCount; i++)
        if ((type_bits & (1u << i)) && (mp.memoryTypes[i].propertyFlags & want) == want) return i;
    return UINT32_MAX;
}

void csky_zc_destroy_image(csky_zc_image *z) {
    if (!z) return;
    if (z->frame) csky_external_frame_release(z->frame);          /* unmaps the HIP view first */
    if (z->semaphore) vkDestroySemaphore(z->device, z->semaphore, NULL);
    if (z->image) vkDestroyImage(z->device, z->image, NULL);
    if (z->memory) vkFreeMemory(z->device, z->memory, NULL);
    free(z);
}

/* Steps 2 + 3 above.  Returns 0 and *out on success; < 0 = a CSKY_ERR_* code (csky_last_error(ctx) has the HIP-side text). */
int csky_zc_create_image(csky_ctx *ctx, VkPhysicalDevice phys, VkDevice device, int width, int height, csky_zc_image **out) {
    VkExternalMemoryImageCreateInfo ext_img = {VK_STRUCTURE_TYPE_EXTERNAL_MEMORY_IMAGE_CREATE_INFO, NULL, VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT};
    VkImageCreateInfo ici;
    VkMemoryRequirements req;
    VkExportMemoryAllocateInfo exp_mem = {VK_STRUCTURE_TYPE_EXPORT_MEMORY_ALLOCATE_INFO, NULL, VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT};
    VkMemoryDedicatedAllocateInfo dedicated = {VK_STRUCTURE_TYPE_MEMORY_DEDICATED_ALLOCATE_INFO, &exp_mem, VK_NULL_HANDLE, VK_NULL_HANDLE};
    VkMemoryAllocateInfo mai = {VK_STRUCTURE_TYPE_MEMORY_ALLOCATE_INFO, &dedicated, 0, 0};
    VkImageSubresource sub = {VK_IMAGE_ASPECT_COLOR_BIT, 0, 0};
    VkSubresourceLayout layout;
    VkMemoryGetFdInfoKHR get_fd = {VK_STRUCTURE_TYPE_MEMORY_GET_FD_INFO_KHR, NULL, VK_NULL_HANDLE, VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT};
    VkExportSemaphoreCreateInfo exp_sem = {VK_STRUCTURE_TYPE_EXPORT_SEMAPHORE_CREATE_INFO, NULL, VK_EXTERNAL_SEMAPHORE_HANDLE_TYPE_OPAQUE_FD_BIT};
    VkSemaphoreCreateInfo sci = {VK_STRUCTURE_TYPE_SEMAPHORE_CREATE_INFO, &exp_sem, 0};
    VkSemaphoreGetFdInfoKHR get_sem_fd = {VK_STRUCTURE_TYPE_SEMAPHORE_GET_FD_INFO_KHR, NULL, VK_NULL_HANDLE, VK_EXTERNAL_SEMAPHORE_HANDLE_TYPE_OPAQUE_FD_BIT};
    PFN_vkGetMemoryFdKHR p_get_memory_fd = (PFN_vkGetMemoryFdKHR)vkGetDeviceProcAddr(device, "vkGetMemoryFdKHR");
    PFN_vkGetSemaphoreFdKHR p_get_semaphore_fd = (PFN_vkGetSemaphoreFdKHR)vkGetDeviceProcAddr(device, "vkGetSemaphoreFdKHR");
    csky_zc_image *z;
    int fd = -1, rc;
    if (!ctx || !out || width < 1 || height < 1 || !p_get_memory_fd || !p_get_semaphore_fd) return CSKY_ERR_INVALID;   /* VK_KHR_external_memory_fd / _semaphore_fd must be enabled */
    *out = NULL;
    z = (csky_zc_image *)calloc(1, sizeof *z);
    if (!z) return CSKY_ERR_INVALID;
    z->device = device; z->width = width; z->height = height;
    memset(&ici, 0, sizeof ici);
    ici.sType = VK_STRUCTURE_TYPE_IMAGE_CREATE_INFO; ici.pNext = &ext_img;
    ici.imageType = VK_IMAGE_TYPE_2D; ici.format = VK_FORMAT_R16G16B16A16_SFLOAT;            /* DATA_FORMAT_R16G16B16A16_SFLOAT, cloud_sky.gd:369 */
    ici.extent.width = (uint32_t)width; ici.extent.height = (uint32_t)height; ici.extent.depth = 1;
    ici.mipLevels = 1; ici.arrayLayers = 1; ici.samples = VK_SAMPLE_COUNT_1_BIT;
    ici.tiling = VK_IMAGE_TILING_LINEAR;                                                      /* rows of pixels: addressable by the HIP kernel */
    ici.usage = VK_IMAGE_USAGE_SAMPLED_BIT | VK_IMAGE_USAGE_TRANSFER_DST_BIT;
    ici.sharingMode = VK_SHARING_MODE_EXCLUSIVE; ici.initialLayout = VK_IMAGE_LAYOUT_UNDEFINED;
    if (vkCreateImage(device, &ici, NULL, &z->image) != VK_SUCCESS) { csky_zc_destroy_image(z); return CSKY_ERR_HIP; }
    vkGetImageMemoryRequirements(device, z->image, &req);
    dedicated.image = z->image;
    mai.allocationSize = req.size;
    mai.memoryTypeIndex = csky_zc_memory_type(phys, req.memoryTypeBits, VK_MEMORY_PROPERTY_DEVICE_LOCAL_BIT);
    if (mai.memoryTypeIndex == UINT32_MAX || vkAllocateMemory(device, &mai, NULL, &z->memory) != VK_SUCCESS ||
        vkBindImageMemory(device, z->image, z->memory, 0) != VK_SUCCESS) { csky_zc_destroy_image(z); return CSKY_ERR_HIP; }
    z->allocation_bytes = req.size;
    vkGetImageSubresourceLayout(device, z->image, &sub, &layout);
    z->row_pitch = layout.rowPitch; z->offset = layout.offset;
    get_fd.memory = z->memory;
    if (p_get_memory_fd(device, &get_fd, &fd) != VK_SUCCESS) { csky_zc_destroy_image(z); return CSKY_ERR_HIP; }
    rc = csky_external_frame_import_fd(ctx, fd, (size_t)z->allocation_bytes, (size_t)z->offset, (size_t)(z->row_pitch * (VkDeviceSize)height), &z->frame, &z->d_ptr);
    if (rc != CSKY_OK) { close(fd); csky_zc_destroy_image(z); return rc; }                    /* on success the HIP runtime owns the fd */
    if (vkCreateSemaphore(device, &sci, NULL, &z->semaphore) != VK_SUCCESS) { csky_zc_destroy_image(z); return CSKY_ERR_HIP; }
    get_sem_fd.semaphore = z->semaphore;
    if (p_get_semaphore_fd(device, &get_sem_fd, &fd) != VK_SUCCESS) { csky_zc_destroy_image(z); return CSKY_ERR_HIP; }
    rc = csky_external_frame_import_semaphore_fd(ctx, z->frame, fd);
    if (rc != CSKY_OK) {                       /* the runtime cannot import semaphores (ROCm 7.2 / Linux): host-side fence instead, step 3 */
        close(fd);
        vkDestroySemaphore(device, z->semaphore, NULL);
        z->semaphore = VK_NULL_HANDLE;
    }
    *out = z;
    return CSKY_OK;
}

/* Step 5: one update pass straight into the image.  `pc` = _fill_push_constant()'s 28 floats; the whole tile in one call (cloud_sky.gd:234-248). */
int csky_zc_render(csky_ctx *ctx, csky_zc_image *z, const csky_cloud_params *pc, void *hip_stream) {
    const csky_bands whole = {z->height, 0, 1, 1};
    int rc = csky_render_clouds_device(ctx, pc, z->width, &whole, z->d_ptr, (size_t)z->row_pitch, hip_stream);
    if (rc != CSKY_OK) return rc;
    if (z->semaphore) return csky_external_frame_signal(ctx, z->frame, hip_stream);      /* the engine's sampling waits on z->semaphore */
    return csky_external_frame_fence(ctx, z->frame, hip_stream);                             /* ... or the host polls csky_zc_ready() */
}
/* 1 = the last csky_zc_render() into this image is complete (safe to sample), 0 = still marching, negative = error. */
int csky_zc_ready(csky_ctx *ctx, csky_zc_image *z) { return z->semaphore ? 1 : csky_external_frame_ready(ctx, z->frame); }
uint64_t csky_zc_image_handle(const csky_zc_image *z) { return (uint64_t)z->image; }       /* -> rd.texture_create_from_extension(..., image, w, h, 1, 1) */
VkSemaphore csky_zc_wait_semaphore(const csky_zc_image *z) { return z->semaphore; }
#endif /* CSKY_HAVE_VULKAN */
